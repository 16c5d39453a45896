"""GPU: the read_label-compatible CLI vs the oracle's whole-file run (FASTA + FASTQ, -t 1 and -t 3)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "lmat_amd", "csrc", "read_label")


def _run_cli(ds, out, query, threads, extra=()):
    args = [EXE, "-f", ds["idmap"], "-u", ds["names"], "-w", ds["rank"], "-x", "0", "-j", "30", "-l", "0", "-b", "1.0",
            "-e", ds["depth"], "-p", "-t", str(threads), "-i", query, "-d", ds["db"], "-c", ds["tree"], "-o", out, *extra]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    return r.stdout


@pytest.mark.parametrize("fastq", [False, True])
def test_cli_matches_oracle_single_shard(small_dataset, tmp_path, fastq):
    import oracle_py
    ds = small_dataset
    out = str(tmp_path / "o")
    q = ds["fastq"] if fastq else ds["fasta"]
    stdout = _run_cli(ds, out, q, 1, ("-q",) if fastq else ())
    o = oracle_py.Oracle(ds["tree"], ds["depth"], ds["rank"], ds["idmap"])
    o.add_taxhisto(ds["db"])
    o.set_options(fastq=int(fastq))
    want, fs, nm = o.run_file(q, 20, ds["names"])
    o.close()
    assert open(out + "0.out").read() == want
    assert open(out + ".0.30.nomatchsum").read() == nm
    assert open(out + ".0.30.fastsummary").read() == fs
    assert "DONE! Total query time" in stdout and "Total reads loaded" in stdout


def test_cli_shards_are_a_partition(small_dataset, tmp_path):
    ds = small_dataset
    _run_cli(ds, str(tmp_path / "a"), ds["fasta"], 1)
    _run_cli(ds, str(tmp_path / "b"), ds["fasta"], 3)
    one = open(str(tmp_path / "a") + "0.out").read()
    three = "".join(open(str(tmp_path / "b") + f"{i}.out").read() for i in range(3))
    assert one == three  # contiguous blocks: concatenation in shard order is the -t 1 file
    assert open(str(tmp_path / "a") + ".0.30.fastsummary").read() == open(str(tmp_path / "b") + ".0.30.fastsummary").read()


def test_cli_refuses_unsupported_and_missing_args(small_dataset, tmp_path):
    ds = small_dataset
    r = subprocess.run([EXE, "-t", "1", "-o", str(tmp_path / "x"), "-i", ds["fasta"]], capture_output=True, text=True)
    assert r.returncode == 255 and "ERROR! Missing depth_file" in r.stderr
    r = subprocess.run([EXE, "-f", ds["idmap"], "-e", ds["depth"], "-t", "1", "-i", ds["fasta"], "-d", ds["db"], "-c", ds["tree"],
                        "-o", str(tmp_path / "x"), "-g", "3", "-m", "no_such_rank_map.txt"], capture_output=True, text=True)
    assert r.returncode != 0 and "rank map" in r.stderr
    r = subprocess.run([EXE, "-f", ds["idmap"], "-e", ds["depth"], "-t", "1", "-i", ds["fasta"], "-d", ds["db"], "-c", ds["tree"],
                        "-o", str(tmp_path / "x"), "-n", "no_such_list.txt"], capture_output=True, text=True)
    assert r.returncode != 0 and "RandHits file list" in r.stderr
