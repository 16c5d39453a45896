"""GPU: the multi-GPU pieces of the path on a one-GPU box -- the multi-rank flow of bench.py (N ranks on one card over
gloo), the engine's own RCCL all-reduce with a one-rank communicator, database replicas (lmat_db_clone), and the CLI's
refusal of input that does not fit its pinned buffers."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "lmat_amd", "csrc", "read_label")


def _engine(ds, params=None, build=True):
    from lmat_amd import Engine, Params
    e = Engine(0, params or Params.run_rl())
    e.load_taxonomy(ds["tree"], ds["depth"], ds["rank"], ds["idmap"])
    if build:
        e.build_db(ds["db"], k=20)
    return e


def _bench(tmp_path, name, *args, env=None):
    out = str(tmp_path / (name + ".tally.json"))
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    base.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--db-gb", "8.3", "--batch", "30000", "--launch-reads", "10000",
                        "--steps", "2", "--warmup", "1", "--windows", "1", "--no-cpu", "--no-e2e", "--tally-out", out, *args],
                       env=base, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, rank 0's
    return json.loads(lines[0]), json.load(open(out))


def test_bench_two_ranks_add_up_to_one_rank_over_the_same_reads(tmp_path):
    """`python bench.py --gpus 2` with no launcher: two rank processes (here both on the one GPU, gloo between them --
    LMAT_BENCH_REHEARSE=1; on a multi-GPU node one per GPU with the engine's RCCL all-reduce).  The line says n_gpus 2, counts
    twice the reads, and the merged tallies equal those of ONE rank that classifies both ranks' read sets itself."""
    two, t2 = _bench(tmp_path, "two", "--gpus", "2", env={"LMAT_BENCH_REHEARSE": "1"})
    one, t1 = _bench(tmp_path, "one", "--gpus", "1", "--as-ranks", "2")
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["reads_per_step_per_gpu"] == 30000 and two["config"]["launches_per_step"] == 3
    assert t2["n_gpus"] == 2
    assert t2["counts"] == t1["counts"] and t2["nomatch"] == t1["nomatch"]
    assert sum(t2["counts"].values()) + sum(t2["nomatch"]) == 2 * 2 * 30000  # ranks x steps x batch: every read tallied once
    for t, s in t1["scores"].items():
        assert abs(t2["scores"][t] - s) <= 1e-9 * max(1.0, abs(s))
    assert 0 < two["roofline"]["frac_step"] <= 1.0 and 0 < two["roofline"]["frac"] <= 1.0
    assert two["value_median_of_windows"] > 0


def test_engine_rccl_allreduce_with_a_one_rank_communicator(small_dataset):
    """lmat_comm_* is the product's RCCL path (collective.cpp: librccl opened on first use, ncclCommInitRank, grouped
    ncclAllReduce of u64 / f64 / u64 on the tally buffer in HBM).  One rank: the sum over ranks is the rank's own tallies."""
    from lmat_amd import Engine
    ds = small_dataset
    e = _engine(ds)
    e.counts_reset()
    e.classify(e.upload_reads(ds["reads"]))
    before = e.counts()
    assert e.lib.lmat_comm_size(e.ctx) == 0
    uid = Engine.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    e.comm_init(uid, 1, 0)
    assert e.lib.lmat_comm_size(e.ctx) == 1
    e.comm_allreduce_counts()
    e.comm_allreduce_counts()
    assert e.counts() == before and sum(c for c, _ in before[0].values()) > 100
    e.lib.lmat_comm_destroy(e.ctx)
    assert e.lib.lmat_comm_size(e.ctx) == 0
    e.close()


def test_allreduce_without_a_communicator_is_an_error(small_dataset):
    from lmat_amd.capi import LmatError
    e = _engine(small_dataset, build=False)
    with pytest.raises(LmatError, match="lmat_comm_init first"):
        e.comm_allreduce_counts()
    e.close()


def test_database_replica_gives_the_same_answers(small_dataset):
    """lmat_db_clone: the second context receives table, overflow table and list arena device to device (what read_label
    does for GPUs 1..N-1 instead of parsing the database N times) and labels reads exactly as the context that built them."""
    from lmat_amd.capi import LmatError
    ds = small_dataset
    a = _engine(ds)
    b = _engine(ds, build=False)
    b.clone_db_from(a)
    assert (b.db_size, b.k, b.table_bytes, b.arena_bytes) == (a.db_size, a.k, a.table_bytes, a.arena_bytes)
    ra, ca = a.classify(a.upload_reads(ds["reads"]))
    rb, cb = b.classify(b.upload_reads(ds["reads"]))
    assert a.format_out(ra, ca) == b.format_out(rb, cb)
    km = np.array([int(x) for x in np.random.default_rng(5).integers(0, 1 << 40, 5000)], dtype=np.uint64)
    assert all((x == y).all() for x, y in zip(a.lookup(km), b.lookup(km)))
    with pytest.raises(LmatError, match="already holds"):
        b.clone_db_from(a)
    a.close()
    rb2, cb2 = b.classify(b.upload_reads(ds["reads"]))  # the replica owns its memory: the source may go away
    assert b.format_out(rb2, cb2) == b.format_out(rb, cb)
    b.close()


def _cli(ds, out, query, extra_env=None):
    args = [EXE, "-f", ds["idmap"], "-u", ds["names"], "-w", ds["rank"], "-x", "0", "-j", "30", "-l", "0", "-b", "1.0",
            "-e", ds["depth"], "-p", "-t", "1", "-i", query, "-d", ds["db"], "-c", ds["tree"], "-o", out]
    return subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, **(extra_env or {})), timeout=600)


def test_cli_fails_loudly_on_a_line_longer_than_its_buffers(small_dataset, tmp_path):
    """A FASTA record on ONE line of 12 MB (a contig) exceeds the 9 MiB pinned batch buffers: the run must end with an error,
    never write past a buffer (the piece cutter only looked at newline positions)."""
    ds = small_dataset
    rng = np.random.default_rng(11)
    q = str(tmp_path / "long.fa")
    with open(q, "w") as f:
        for i in range(50):
            f.write(">r%d\n%s\n" % (i, ds["reads"][i]))
        f.write(">contig\n" + "".join(rng.choice(list("ACGT"), 12 << 20)) + "\n")
        for i in range(50, 80):
            f.write(">r%d\n%s\n" % (i, ds["reads"][i]))
    r = _cli(ds, str(tmp_path / "o"), q)
    assert r.returncode != 0 and "exceeds the batch buffer" in r.stderr, r.stderr[-2000:]
    # the same record folded into 60-column lines fits a buffer piece by piece, but is far beyond the engine's longest read:
    # refused by the engine (never cut into two reads silently)
    q2 = str(tmp_path / "folded.fa")
    with open(q2, "w") as f:
        s = "".join(rng.choice(list("ACGT"), 10 << 20))
        f.write(">contig\n" + "\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n")
        f.write(">r1\n%s\n" % ds["reads"][1])
    r = _cli(ds, str(tmp_path / "o2"), q2)
    assert r.returncode != 0 and "ERROR" in r.stderr, r.stderr[-2000:]
    # last line without a newline, longer than a piece but within a buffer: still fine
    q3 = str(tmp_path / "tail.fa")
    with open(q3, "w") as f:
        for i in range(40):
            f.write(">r%d\n%s\n" % (i, ds["reads"][i]))
        f.write(">last\n" + ds["reads"][41])
    r = _cli(ds, str(tmp_path / "o3"), q3, {"LMAT_FASTA_PIECE": "2000"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(open(str(tmp_path / "o3") + "0.out").read().splitlines()) == 41


def test_cli_with_two_contexts_clones_the_database(small_dataset, tmp_path):
    """LMAT_DEVICES=0,0: the second context gets its database from the first (lmat_db_clone) and both feed the writer."""
    ds = small_dataset
    a = _cli(ds, str(tmp_path / "a"), ds["fasta"], {"LMAT_DEVICES": "0", "LMAT_FASTA_PIECE": "4000"})
    b = _cli(ds, str(tmp_path / "b"), ds["fasta"], {"LMAT_DEVICES": "0,0", "LMAT_FASTA_PIECE": "4000"})
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    for suffix in ("0.out", ".0.30.fastsummary", ".0.30.nomatchsum"):
        assert open(str(tmp_path / "a") + suffix).read() == open(str(tmp_path / "b") + suffix).read(), suffix


def test_cli_writes_the_rollups_itself(small_dataset, tmp_path):
    """LMAT_ROLLUPS=<ranks>: read_label writes <fastsummary>.lineage and <fastsummary>.<rank> after the summaries -- what
    bin/run_rl.sh:251-252 runs tolineage.py / fsreport.py for -- from the inputs the run already has; the files equal what the
    stand-alone tool makes of the same .fastsummary."""
    ds = small_dataset
    r = _cli(ds, str(tmp_path / "o"), ds["fasta"], {"LMAT_ROLLUPS": "species,genus"})
    assert r.returncode == 0, r.stderr
    fs = str(tmp_path / "o") + ".0.30.fastsummary"
    got = {s: open(fs + s).read() for s in (".lineage", ".species", ".genus")}
    assert got[".species"].count("\n") > 3   # (.lineage lists calls with more than 10 reads, as run_rl.sh asks: may be empty here)
    for s in got:
        os.rename(fs + s, fs + s + ".cli")
    subprocess.run([os.path.join(ROOT, "lmat_amd", "csrc", "fs_rollup"), "-s", fs, "-u", ds["names"], "-c", ds["tree"], "-w", ds["rank"],
                    "-a", "species,genus"], check=True)
    for s in got:
        assert open(fs + s).read() == got[s], s
    # every read called at or below a species is in exactly one species row
    total = sum(int(l.split("\t")[2]) for l in got[".species"].splitlines()[1:])
    called = sum(int(l.split("\t")[1]) for l in open(fs))
    assert 0 < total <= called
