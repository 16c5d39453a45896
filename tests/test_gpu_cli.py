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


def test_cli_with_several_contexts_writes_the_same_files(small_dataset, tmp_path, monkeypatch):
    """read_label drives one context per GPU (src/read_label.cpp:1637-1800 is its OpenMP fan-out + merge); here two
    contexts share the one GPU of the box (LMAT_DEVICES=0,0), tiny FASTA pieces make many batches so that both get
    work, and the writer restores input order: the .out shards and both summaries equal the single-context run."""
    ds = small_dataset
    monkeypatch.setenv("LMAT_FASTA_PIECE", "4000")
    monkeypatch.setenv("LMAT_DEVICES", "0")
    _run_cli(ds, str(tmp_path / "a"), ds["fasta"], 2)
    monkeypatch.setenv("LMAT_DEVICES", "0,0")
    out = _run_cli(ds, str(tmp_path / "b"), ds["fasta"], 2)
    assert "on each of 2 GPUs" in out
    for suffix in ("0.out", "1.out", ".0.30.fastsummary", ".0.30.nomatchsum"):
        assert open(str(tmp_path / "a") + suffix).read() == open(str(tmp_path / "b") + suffix).read(), suffix
    assert len(open(str(tmp_path / "b") + "0.out").read()) > 10000


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


def test_rand_read_label_cli(tmp_path):
    """The null-model generator's command line: random reads in ten GC buckets against a 12-mer database (so random
    reads do hit); the .rand_lst rows equal the oracle's restatement of rand_read_label on the dumped reads."""
    import numpy as np
    import oracle_py
    from lmat_amd import synth
    tax = synth.make_taxonomy((2, 2, 2, 2, 3, 3), specials=True)
    p = synth.write_aux_files(str(tmp_path), tax)
    genomes = synth.make_genomes(tax, 2000, 2002)
    kmers, lists = synth.build_kmer_table(tax, genomes, 12)
    p["db"] = os.path.join(str(tmp_path), "th.bin")
    synth.write_taxhisto(p["db"], kmers, lists, 12)
    out = os.path.join(str(tmp_path), "nm")
    dump = os.path.join(str(tmp_path), "reads.fa")
    exe = os.path.join(ROOT, "lmat_amd", "csrc", "rand_read_label")
    r = subprocess.run([exe, "-d", p["db"], "-c", p["tree"], "-e", p["depth"], "-f", p["idmap"], "-w", p["rank"], "-t", "2", "-g", "1500",
                        "-i", "80", "-o", out, "-S", "42", "-O", dump], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Total reads to evaluate: 3000" in r.stdout
    reads, gc = [], []
    for line in open(dump):
        if line.startswith(">"):
            gc.append(int(line.split("gc=")[1]))
        else:
            reads.append(line.strip())
    assert len(reads) == 3000 and all(len(x) == 80 for x in reads)
    # bucket b holds reads with b*10 .. b*10+9 percent G+C (genRandRead)
    for x, b in zip(reads[:200], gc[:200]):
        frac = sum(ch in "gc" for ch in x) / 80.0
        assert b * 0.10 - 0.02 <= frac <= b * 0.10 + 0.10
    bs = [x.encode() for x in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    orc = oracle_py.Oracle(p["tree"], p["depth"], p["rank"], p["idmap"])
    orc.add_taxhisto(p["db"])
    want = orc.rand_label(np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8), off, 12, np.array(gc, dtype=np.uint8))
    orc.close()
    rows = [l.split() for l in open(out + ".rand_lst")]
    assert len(rows) == len(want) > 20
    assert [int(x[0]) for x in rows] == sorted(want)
    for x in rows:
        mx, ct = want[int(x[0])]
        assert len(x) == 21
        for b in range(10):
            assert x[1 + 2 * b] == "%g" % float(mx[b]) and int(x[2 + 2 * b]) == int(ct[b])


def test_parallel_fasta_front_end_equals_the_sequential_reader(small_dataset, tmp_path):
    """A FASTA file is mapped and parsed in pieces in parallel; stdin goes through the sequential reader that
    restates the reference's producer loop.  Same text either way, also for a file full of the reader's corner
    cases (multi-line records, blank and one-character lines, records without sequence or header, no final newline)
    and with pieces of a few hundred bytes so that many cuts fall inside the file."""
    ds = small_dataset
    real = [l.rstrip("\n") for l in open(ds["fasta"])]
    tricky = str(tmp_path / "tricky.fa")
    with open(tricky, "w") as f:
        f.write("ACGTACGTACGTACGTACGTACGTACGTACGTAC\n")            # sequence before any header
        f.write(">multi line record\n" + real[1][:70] + "\n\n" + real[1][70:] + "\nA\n")
        f.write(">\n" + real[3] + "\n")                              # empty header
        f.write(">no sequence at all\n>another without\n\n")
        f.write(">x y z\tw\n" + real[5] + "\r\n")                    # CR kept as data, tab in the header
        for i in range(7, 201, 2):
            f.write(real[i - 1] + "\n" + "\n".join(real[i][j:j + 60] for j in range(0, len(real[i]), 60)) + "\n")
        f.write(">last one without newline\n" + real[9])
    for name, query in (("plain", ds["fasta"]), ("tricky", tricky)):
        outs = []
        for mode in ("file", "stdin", "pieces"):
            out = str(tmp_path / f"{name}_{mode}")
            args = [EXE, "-f", ds["idmap"], "-u", ds["names"], "-w", ds["rank"], "-x", "0", "-j", "30", "-l", "0", "-b", "1.0",
                    "-e", ds["depth"], "-p", "-t", "1", "-i", "-" if mode == "stdin" else query, "-d", ds["db"], "-c", ds["tree"], "-o", out]
            env = dict(os.environ, LMAT_FASTA_PIECE="300") if mode == "pieces" else dict(os.environ)
            r = subprocess.run(args, capture_output=True, text=True, env=env, stdin=open(query) if mode == "stdin" else None)
            assert r.returncode == 0, r.stderr + r.stdout
            outs.append(open(out + "0.out").read() + open(out + ".0.30.fastsummary").read())  # (-t 1: shards cut every batch)
        assert outs[0] == outs[1] == outs[2], name
        assert "unknown_hdr:" in outs[0] or name == "plain"


def test_parallel_fastq_front_end_equals_the_sequential_reader(small_dataset, tmp_path):
    """A FASTQ file is mapped and parsed in pieces too (round 4): pieces start where the sequential reader is provably between
    two records, and run its state machine -- header of the PREVIOUS record, one quality line skipped (read_label.cpp:1651-1713).
    Same files as the one-thread reader (LMAT_FASTQ_SEQUENTIAL=1) and as stdin, with pieces of a few hundred bytes, for the
    plain file and for one full of traps: quality lines that start with '@', '+' and '-', '-' separators, a '>' line, blank
    lines, CR LF, no final newline."""
    ds = small_dataset
    real = [l.rstrip("\n") for l in open(ds["fasta"]) if not l.startswith(">")]
    tricky = str(tmp_path / "tricky.fq")
    with open(tricky, "w") as f:
        for i, s_ in enumerate(real[:240]):
            q = "I" * len(s_)
            if i % 7 == 1: q = "@" + q[1:]          # a quality line that looks like a header
            if i % 7 == 2: q = "+" + q[1:]          # ... like a separator
            if i % 7 == 3: q = "-" + q[1:]
            sep = "-" if i % 5 == 4 else ("+r%d" % i if i % 5 == 3 else "+")
            if i == 20: f.write(">stray fasta header\n")   # header AND sequence in -q mode
            if i == 30: f.write("\n")
            eol = "\r\n" if i % 11 == 5 else "\n"
            f.write("@q%d some text\t%d%s%s%s%s%s%s" % (i, i, eol, s_, eol, sep, eol, q))
            f.write(eol if i < 239 else "")
    for name, query in (("plain", ds["fastq"]), ("tricky", tricky)):
        for shards in (1,):
            outs = []
            for mode in ("sequential", "stdin", "pieces", "pieces_big"):
                out = str(tmp_path / f"{name}_{shards}_{mode}")
                args = [EXE, "-f", ds["idmap"], "-u", ds["names"], "-w", ds["rank"], "-x", "0", "-j", "30", "-l", "0", "-b", "1.0", "-q",
                        "-e", ds["depth"], "-p", "-t", str(shards), "-i", "-" if mode == "stdin" else query, "-d", ds["db"], "-c", ds["tree"], "-o", out]
                env = dict(os.environ)
                if mode == "sequential": env["LMAT_FASTQ_SEQUENTIAL"] = "1"
                if mode == "pieces": env["LMAT_FASTA_PIECE"] = "700"
                r = subprocess.run(args, capture_output=True, text=True, env=env, stdin=open(query) if mode == "stdin" else None)
                assert r.returncode == 0, r.stderr + r.stdout
                # (shards cut every batch: with several shards the shard files depend on the batching, their sorted lines do not)
                body = "".join(open(out + f"{t}.out").read() for t in range(shards))
                outs.append((sorted(body.splitlines()) if shards > 1 else body, open(out + ".0.30.fastsummary").read()))
            assert outs[0] == outs[1] == outs[2] == outs[3], (name, shards)
            text = outs[0][0] if shards == 1 else "\n".join(outs[0][0])
            assert "unknown_hdr:1\t" in text


# ---- the contract with bin/run_rl.sh: its own argv, recorded from the script (tests/golden/make_run_rl_argv.py) ----------
def _run_rl_cases():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "run_rl_argv.json")))


@pytest.mark.parametrize("case", [c["name"] for c in _run_rl_cases()])
def test_cli_accepts_the_argv_run_rl_sh_builds(tmp_path, case):
    """read_label must keep working unmodified under bin/run_rl.sh.  The argv of each case is what the
    reference's script itself passed to a recording stub (bin/run_rl.sh:88-243); here $LMAT_DIR is filled with a synthetic
    data set under the file names the script hard-codes, the argv is run verbatim, and the three outputs must be the
    oracle's for the options the argv spells."""
    import shutil
    import oracle_py
    from lmat_amd import synth
    c = [x for x in _run_rl_cases() if x["name"] == case][0]
    lmat_dir, odir = tmp_path / "lmatdir", tmp_path / "out"
    lmat_dir.mkdir()
    odir.mkdir()
    info = synth.generate_dataset(str(tmp_path / "ds"), (2, 2, 2, 2, 3, 3), 1500, 1200, L=(60, 100, 150, 250), frac_short=0.02)
    tax = synth.make_taxonomy((2, 2, 2, 2, 3, 3), True)
    names = {"tree": "ncbi_taxonomy.segment.pruned.dat.nohl", "depth": "depth_for_ncbi_taxonomy.segment.pruned.dat",
             "rank": "ncbi_taxid_to_rank.pruned.txt", "names": "ncbi_taxonomy_rank.segment.pruned.txt", "idmap": "m9.32To16.map"}
    for key, fn in names.items():
        shutil.copy(info[key], lmat_dir / fn)
    with open(lmat_dir / "numeric_ranks", "w") as f:
        for t in tax.ids:
            f.write(f"{t} {tax.depth[t]}\n")
    db = tmp_path / "kML.test.db"
    shutil.copy(info["db"], db)
    nm_lst = synth.write_null_models(str(lmat_dir), tax)                      # gz tables + a list, paths relative to $LMAT_DIR
    shutil.copy(nm_lst, lmat_dir / (db.name + ".null_lst.txt"))
    shutil.copy(nm_lst, lmat_dir / "my_null_lst.txt")
    query = tmp_path / ("reads.fq" if c["fastq"] else "reads.fna")
    shutil.copy(info["fastq"] if c["fastq"] else info["fasta"], query)
    sub = {"$LMAT_DIR": str(lmat_dir), "$ODIR": str(odir), "$QUERYNAME": query.name, "$QUERY": str(query), "$DBNAME": db.name, "$DB": str(db)}

    def fill(s):
        for k in ("$LMAT_DIR", "$ODIR", "$QUERYNAME", "$QUERY", "$DBNAME", "$DB"):
            s = s.replace(k, sub[k])
        return s
    argv = [fill(a) for a in c["read_label_argv"]]
    r = subprocess.run([EXE] + argv, capture_output=True, text=True, env=dict(os.environ, LMAT_DIR=str(lmat_dir)))
    assert r.returncode == 0, r.stderr + r.stdout
    # the oracle with the options this argv spells (getopt letters of src/read_label.cpp:1351-1441)
    opt = {argv[i]: argv[i + 1] for i in range(len(argv) - 1) if argv[i].startswith("-") and not argv[i + 1].startswith("-")}
    os.environ["LMAT_DIR"] = str(lmat_dir)
    o = oracle_py.Oracle(fill("$LMAT_DIR/" + names["tree"]), fill("$LMAT_DIR/" + names["depth"]), fill("$LMAT_DIR/" + names["rank"]), opt["-f"])
    if "-g" in opt:
        o.set_label_modes(False, int(opt["-g"]), opt.get("-m"))
    o.add_taxhisto(str(db))
    o.set_options(sdiff=float(opt["-b"]), hbias=float(opt["-l"]), prn_all=int("-p" in argv), min_score=float(opt["-x"]),
                  min_kmer=int(opt["-j"]), fastq=int("-q" in argv))
    if "-n" in opt:
        o.load_null_models(opt["-n"])
    want, fs, nm = o.run_file(str(query), 20, opt["-u"])
    o.close()
    nshard = int(opt["-t"])
    got = "".join(open(opt["-o"] + f"{i}.out").read() for i in range(nshard))
    assert got == want
    sbase = "%s.%s.%s" % (opt["-o"], opt["-x"], opt["-j"])
    assert open(sbase + ".fastsummary").read() == fs
    assert open(sbase + ".nomatchsum").read() == nm
    assert len(want) > 50000


def test_cli_with_a_taxonomy_above_16_bits(small_dataset, tmp_path):
    """A taxonomy of more than 65534 nodes (70 000 unrelated nodes added under the root) with the 32->16 map that
    make_db_image -M derives from the database's own taxids: read_label writes the files of the run on the plain taxonomy."""
    ds = small_dataset
    _run_cli(ds, str(tmp_path / "a"), ds["fasta"], 1)
    big = str(tmp_path / "tax_big.dat")
    lines = [l for l in open(ds["tree"]).read().split("\n")]
    with open(big, "w") as f:
        f.write("\n".join(lines[:3]) + "\n" + "\n".join(l for l in lines[3:] if l != "") + "\n")
        for i in range(70000):
            f.write("%d 0 1\nfiller node %d\n" % (900000000 + i, i))
    mp, img = str(tmp_path / "derived_map.txt"), str(tmp_path / "db.img")
    r = subprocess.run([os.path.join(ROOT, "lmat_amd", "csrc", "make_db_image"), "-i", ds["db"], "-o", img, "-k", "20", "-t", big, "-M", mp],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    big_ds = dict(ds, tree=big, idmap=mp, db=img)
    _run_cli(big_ds, str(tmp_path / "b"), ds["fasta"], 1)
    for suffix in ("0.out", ".0.30.fastsummary", ".0.30.nomatchsum"):
        assert open(str(tmp_path / "a") + suffix).read() == open(str(tmp_path / "b") + suffix).read(), suffix
