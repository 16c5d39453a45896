"""Randomised differential test: many small seeded configurations (taxonomy shape, genome size, k, read lengths and
error mix, scoring parameters, label modes) -- HIP path through the C-ABI vs the CPU oracle, byte-identical text."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _blob(reads):
    bs = [r.encode() for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    return np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8), off


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("LMAT_FUZZ_SEEDS", "32")))))  # LMAT_FUZZ_SEEDS=N for a longer soak
def test_random_configuration_matches_oracle(seed, tmp_path):
    from lmat_amd import Engine, Params, synth
    import oracle_py
    rng = np.random.default_rng(1000 + seed)
    branching = tuple(int(rng.integers(1, 4)) for _ in range(4)) + (int(rng.integers(2, 4)), int(rng.integers(1, 5)))
    k = int(rng.choice([16, 18, 20, 20]))
    G = int(rng.integers(300, 1500))
    pool = [40, 75, 100, 150, 151, 179, 180, 250, 300, 531, 600]
    if os.environ.get("LMAT_FUZZ_LONG"):  # soak option: contig-length reads reach the 2048-k-mer and the global-memory classes
        pool += [1500, 2067, 2068, 5000]
    lens = [int(x) for x in rng.choice(pool, size=4)]
    tax = synth.make_taxonomy(branching, specials=bool(rng.integers(0, 2)))
    p = synth.write_aux_files(str(tmp_path), tax)
    genomes = synth.make_genomes(tax, G, 2002 + seed, strain_sub=float(rng.choice([0.002, 0.01, 0.05])),
                                 genus_block=float(rng.choice([0.0, 0.1, 0.4])))
    kmers, lists = synth.build_kmer_table(tax, genomes, k)
    p["db"] = os.path.join(str(tmp_path), "th.bin")
    synth.write_taxhisto(p["db"], kmers, lists, k)
    reads = synth.make_reads(tax, genomes, 300, lens, 3003 + seed, err=float(rng.choice([0.0, 0.01, 0.05])),
                             frac_random=0.1, frac_n=0.05, frac_lowc=0.03, frac_short=0.03, lower_frac=0.1)
    seqs = [r for _, r in reads]
    opts = dict(sdiff=float(rng.choice([0.5, 1.0, 2.0])), hbias=float(rng.choice([0.0, 3.0])), prn_all=int(rng.integers(0, 2)),
                screen_phix=int(rng.integers(0, 2)), min_score=float(rng.choice([0.0, 0.3])), min_kmer=int(rng.choice([1, 30, 35])),
                min_fnd_kmer=int(rng.choice([1, 3])))
    permissive = int(rng.integers(0, 4) == 0)
    use_map = bool(rng.integers(0, 4))  # one in four: database of 32-bit taxids, codes from the tree
    null_models = k == 20 and int(rng.integers(0, 4)) == 0  # one in four of the 20-mer cases: log-odds scores (-n)
    null_lst = None
    if null_models:
        null_lst = synth.write_null_models(os.path.join(str(tmp_path), "nm"), tax, seed=4004 + seed)
        os.environ["LMAT_DIR"] = os.path.join(str(tmp_path), "nm")
    idmap = p["idmap"] if use_map else None
    eng = Engine(0, Params(opts["sdiff"], opts["hbias"], opts["min_score"], opts["min_kmer"], opts["min_fnd_kmer"], opts["prn_all"],
                           opts["screen_phix"]))
    eng.load_taxonomy(p["tree"], p["depth"], p["rank"], idmap)
    if permissive:
        eng.set_label_modes(permissive=1)
    eng.build_db(p["db"], k=k)
    if null_lst:
        eng.load_null_models(null_lst)
    orc = oracle_py.Oracle(p["tree"], p["depth"], p["rank"], idmap)
    if permissive:
        orc.set_label_modes(permissive=1)
    orc.add_taxhisto(p["db"])
    orc.set_options(**opts)
    if null_lst:
        orc.load_null_models(null_lst)
    blob, off = _blob(seqs)
    dr = eng.upload_reads((blob, off))
    res, cands = eng.classify(dr, cand_cap=512 * len(seqs))
    got = eng.format_out(res, cands, (blob, off), 0)
    want, tally, nm = orc.classify(blob, off, k, 0)
    if got != want:
        g, w = got.split("\n"), want.split("\n")
        bad = [(i, a, b) for i, (a, b) in enumerate(zip(g, w)) if a != b]
        raise AssertionError(f"seed {seed} ({branching}, k={k}, {opts}, permissive={permissive}, null_models={null_models}): {len(bad)} records differ; first: {bad[0]}")
    counts, nomatch = eng.counts()
    assert nomatch == nm
    assert {t: c for t, (c, s) in counts.items()} == {t: c for t, (c, s) in tally.items()}
    dr.free()
    eng.close()
    orc.close()
