"""GPU: a taxonomy of more than 65534 ids (upstream's TID_SIZE=32 build, CMakeLists.txt:92-105).  The
database stores 32-bit taxids as they are, internal ids and Euler ticks are 32 bits wide, and the reads run in the WIDE
classes of the classify kernel (kernels.hip: WIDE -- the classes with in-kernel decision, templated on the id width)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wide_dataset(tmp_path_factory):
    """67 584 strains under 1024 species (70 k nodes in all): every strain owns k-mers, so the closure of the database
    -- the ids it can produce plus their ancestors -- is the whole tree, beyond any 16-bit numbering."""
    from lmat_amd import synth
    d = str(tmp_path_factory.mktemp("ds_wide"))
    tax = synth.make_taxonomy((2, 2, 4, 8, 8, 66), specials=True)
    p = synth.write_aux_files(d, tax)
    genomes = synth.make_genomes(tax, 170, 2002, strain_sub=0.004)
    kmers, lists = synth.build_kmer_table(tax, genomes, 20)
    p["db"] = os.path.join(d, "th.bin")
    synth.write_taxhisto(p["db"], kmers, lists, 20)
    reads = synth.make_reads(tax, genomes, 1500, 150, 3003)
    p["reads"] = [r for _, r in reads] if isinstance(reads[0], tuple) else reads
    p["n_nodes"] = len(tax.ids)
    p["kmers"], p["lists"] = kmers, lists
    assert p["n_nodes"] > 65534 and len(set(x for l in lists for x in l)) > 65534
    return p


def test_wide_taxonomy_text_parity(wide_dataset):
    """Byte-identical .out text and tallies against the CPU oracle (which works in 32-bit taxids throughout) on a database
    whose closure exceeds 65 534 ids; the lookups return the stored 32-bit lists."""
    import oracle_py
    from lmat_amd import Engine, Params
    ds = wide_dataset
    eng = Engine(0, Params.run_rl())
    eng.load_taxonomy(ds["tree"], ds["depth"], ds["rank"], None)     # no -f map: 32-bit taxids, a tree beyond 16-bit codes
    eng.build_db(ds["db"], k=20)
    assert eng.db_size == ds["kmers"].size
    # lookups: the stored lists, as 32-bit taxids
    rng = np.random.default_rng(3)
    pick = rng.integers(0, ds["kmers"].size, 3000)
    cnt, tids = eng.lookup(ds["kmers"][pick], stride=160)
    for j, i in enumerate(pick.tolist()):
        want = ds["lists"][i]
        assert cnt[j] == len(want) and tids[j, :len(want)].tolist() == list(want), i
    assert max(int(t) for l in (ds["lists"][i] for i in pick.tolist()) for t in l) > 400000   # ids far beyond 16 bits occur
    reads = ds["reads"]
    bs = [r.encode() for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    blob = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8)
    dr = eng.upload_reads((blob, off))
    eng.counts_reset()
    res, cands = eng.classify(dr, cand_cap=200 * len(reads))
    got = eng.format_out(res, cands, (blob, off))
    counts, nomatch = eng.counts()
    orc = oracle_py.Oracle(ds["tree"], ds["depth"], ds["rank"], None)
    orc.add_taxhisto(ds["db"])
    orc.set_options()
    want, tally, nm = orc.classify(blob, off, 20)
    orc.close()
    assert got == want
    assert (res["status"] == 0).sum() > 0.5 * len(reads)
    called = {int(t) for t in res["call_tid"][res["status"] == 0]}
    assert max(called) > 65535                                      # calls on taxids beyond 16 bits
    assert {t: c for t, (c, _) in counts.items()} == {int(t): int(c) for t, (c, _) in tally.items()} and list(nomatch) == [int(x) for x in nm]
    dr.free()
    eng.close()


def test_wide_taxonomy_cli_and_image(wide_dataset, tmp_path):
    """read_label without -f on the same database: the files equal the oracle's whole-file run; the database survives a
    device-image round trip (LMATIMG2 carries the wide list records as they are)."""
    import subprocess
    import oracle_py
    ds = wide_dataset
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "lmat_amd", "csrc", "read_label")
    fa = str(tmp_path / "q.fa")
    with open(fa, "w") as f:
        for i, r in enumerate(ds["reads"][:600]):
            f.write(">r%d\n%s\n" % (i, r))
    base = [exe, "-u", ds["names"], "-w", ds["rank"], "-x", "0", "-j", "30", "-l", "0", "-b", "1.0", "-e", ds["depth"], "-p", "-t", "1",
            "-i", fa, "-c", ds["tree"]]
    img = str(tmp_path / "wide.img2")
    r1 = subprocess.run(base + ["-d", ds["db"], "-o", str(tmp_path / "o1")], env=dict(os.environ, LMAT_SAVE_DEVICE_IMAGE=img), capture_output=True, text=True)
    assert r1.returncode == 0, r1.stderr + r1.stdout
    r2 = subprocess.run(base + ["-d", img, "-o", str(tmp_path / "o2")], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr + r2.stdout
    orc = oracle_py.Oracle(ds["tree"], ds["depth"], ds["rank"], None)
    orc.add_taxhisto(ds["db"])
    orc.set_options()
    want, fs, nm = orc.run_file(fa, 20, ds["names"])
    orc.close()
    for o in ("o1", "o2"):
        assert open(str(tmp_path / o) + "0.out").read() == want
        assert open(str(tmp_path / o) + ".0.30.fastsummary").read() == fs
        assert open(str(tmp_path / o) + ".0.30.nomatchsum").read() == nm


@pytest.mark.parametrize("mode", ["permissive", "prune", "null_bias"])
def test_wide_taxonomy_label_modes(wide_dataset, tmp_path, mode):
    """The same parity under -s (permissive match), under run-time pruning (-g 6 -m ranks) and with -l 3 / no -p: the wide
    classes are the classes of the general path, so every mode the 16-bit build has is there."""
    import oracle_py
    from lmat_amd import Engine, Params
    ds = wide_dataset
    reads = ds["reads"][:700]
    bs = [r.encode() for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    blob = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8)
    rank_fn = str(tmp_path / "numeric_ranks.txt")
    with open(rank_fn, "w") as f:
        for line in open(ds["depth"]):
            f.write(line)
    prm = Params.run_rl(prn_all=0) if mode == "null_bias" else Params.run_rl()
    if mode == "null_bias":
        prm.hbias = 3.0
    eng = Engine(0, prm)
    eng.load_taxonomy(ds["tree"], ds["depth"], ds["rank"], None)
    orc = oracle_py.Oracle(ds["tree"], ds["depth"], ds["rank"], None)
    if mode == "permissive":
        eng.set_label_modes(permissive=True)
        orc.set_label_modes(True, 0, None)
    elif mode == "prune":
        eng.set_label_modes(False, 6, rank_fn)
        orc.set_label_modes(False, 6, rank_fn)
    eng.build_db(ds["db"], k=20)
    orc.add_taxhisto(ds["db"])
    orc.set_options(hbias=3.0, prn_all=0) if mode == "null_bias" else orc.set_options()
    dr = eng.upload_reads((blob, off))
    res, cands = eng.classify(dr, cand_cap=400 * len(reads))
    got = eng.format_out(res, cands, (blob, off))
    want, _, _ = orc.classify(blob, off, 20)
    assert got == want
    assert (res["status"] == 0).sum() > 0.4 * len(reads)
    orc.close()
    dr.free()
    eng.close()
