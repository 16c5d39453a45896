"""GPU parity: the HIP path through the C-ABI vs the CPU oracle on identical inputs (bit-exact text)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(small_dataset):
    from lmat_amd import Engine, Params
    e = Engine(0, Params.run_rl())
    e.load_taxonomy(small_dataset["tree"], small_dataset["depth"], small_dataset["rank"], small_dataset["idmap"])
    e.build_db(small_dataset["db"], k=20)
    yield e
    e.close()


def _blob(reads):
    bs = [r.encode() for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    return np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8), off


def test_db_size(engine, small_dataset):
    assert engine.db_size == small_dataset["n_kmers"]
    assert engine.k == 20


def test_lookup_matches_oracle(engine, oracle_small, small_dataset):
    kms = []
    for r in small_dataset["reads"]:
        km, _, _, _ = oracle_small.extract(r.encode(), 20)
        kms.append(km)
    kms = np.unique(np.concatenate(kms))
    counts, tids = engine.lookup(kms, stride=32)
    hits = 0
    for i, km in enumerate(kms.tolist()):
        n, lst = oracle_small.lookup(km)
        if n < 0:
            assert counts[i] == 0
        else:
            hits += 1
            assert counts[i] == n
            assert tids[i, :n].tolist() == lst.tolist()
    assert hits > 1000


def test_out_text_matches_oracle(engine, oracle_small, small_dataset):
    reads = small_dataset["reads"]
    blob, off = _blob(reads)
    dr = engine.upload_reads((blob, off))
    engine.counts_reset()
    res, cands = engine.classify(dr)
    text = engine.format_out(res, cands, (blob, off))
    want, tally, nomatch = oracle_small.classify(blob, off, 20)
    got_lines, want_lines = text.split("\n"), want.split("\n")
    bad = [(i, g, w) for i, (g, w) in enumerate(zip(got_lines, want_lines)) if g != w]
    assert not bad, f"{len(bad)} differing records, first: {bad[0]}"
    assert text == want
    counts, nm = engine.counts()
    assert nm == nomatch
    assert {t: c for t, (c, s) in counts.items()} == {t: c for t, (c, s) in tally.items()}
    for t, (c, s) in counts.items():
        assert abs(s - tally[t][1]) <= 1e-4 * max(1.0, abs(s))
    dr.free()


def test_packed_reads_roundtrip(engine, small_dataset):
    reads = small_dataset["reads"][:50]
    blob, off = _blob(reads)
    dr = engine.upload_reads((blob, off))
    b2, o2 = dr.ascii()
    assert o2.tolist() == off.tolist()
    want = bytes(blob[:-1]).upper()
    want = bytes(c if c in b"ACGT" else ord("N") for c in want)
    assert bytes(b2) == want
    dr.free()
