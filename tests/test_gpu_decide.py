"""GPU: the decision kernels against the records the reference's own example run printed (not only against the oracle)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_decision_kernels_reproduce_the_reference_example_calls(tmp_path):
    """593 records of the reference's example output whose taxonomy the run's own .summ report gives
    (tests/golden/example_records.json + example_tree.json): the printed candidates (taxid, null-model score) and the printed
    standard deviation go into the HIP decision code through lmat_debug_decide -- std::sort(TCmp) and findReadLabelVer2 as
    every read of a null-model run goes through them (k4_part1 / k4_part2) --, and out come the call taxid and score the
    REFERENCE printed, on every record (read_label.cpp:284-419, 475-485, 892-896).  The match type equals the CPU oracle's
    on every record (and the reference's wherever the printed text can decide it: see the CPU test of the same name)."""
    import oracle_py
    from lmat_amd import Engine, Params
    from test_oracle_golden import _example_tree_files, example_replay_cases
    files, ids, parent, depth = _example_tree_files(tmp_path)
    cases = example_replay_cases()
    assert len(cases) == 593
    tables = [[(int(t), np.float32(s)) for t, s in r["cands"]] for _, r in cases]
    stdevs = [np.float32(r["stats"][1]) for _, r in cases]
    eng = Engine(0, Params.run_rl())       # -b 1.0 -l 0, as bin/run_rl.sh passes them
    eng.load_taxonomy(*files)
    res = eng.debug_decide(tables, stdevs)
    orc = oracle_py.Oracle(*files)
    orc.set_options(sdiff=1.0, hbias=0.0)
    multi = cross = 0
    for (run, r), got, tb, sd in zip(cases, res, tables, stdevs):
        assert str(int(got["call_tid"])) == r["call"][0], (r["hdr"], got)
        assert abs(float(got["call_score"]) - float(r["call"][1])) < 1e-5, (r["hdr"], got)
        ct, cs, m = orc.replay_decision([t for t, _ in tb], [s for _, s in tb], float(sd))
        assert (int(got["call_tid"]), int(got["match_type"])) == (ct, m) and np.float32(cs) == got["call_score"], r["hdr"]
        multi += m != 0
        tops = set()
        for t, _ in r["cands"]:
            while depth[t] > 1:
                t = parent[t]
            tops.add(t)
        cross += len(tops) > 1
    orc.close()
    assert multi >= 8 and cross >= 10   # poisoned lineages (MultiMatch by the replay itself), and candidate sets that span kingdoms, occur
    # shuffled input order: the sort is part of what is tested
    rng = np.random.default_rng(3)
    shuf = [[tb[i] for i in rng.permutation(len(tb))] for tb in tables]
    res2 = eng.debug_decide(shuf, stdevs)
    same = sum(int(a["call_tid"]) == int(b["call_tid"]) and a["call_score"] == b["call_score"] for a, b in zip(res, res2))
    assert same >= 590   # (TCmp is not a strict weak order: a handful of tables within its 0.001 band may sort differently)
    eng.close()


def _random_count_tables(tax, n_tables, seed):
    """(tids, counts, off, cand) of n_tables candidate tables over the taxonomy: a lineage (a strain and some of its ancestors),
    relatives (strains of the same genus: equal depths, often equal counts -- TCmp ties), strays from anywhere in the tree, now
    and then PhiX; 1 .. 64 distinct taxids per table in random order, counts in a few distinct values so that equal keys are
    common, a third of the tables above 16 entries (std::sort's partition steps)."""
    rng = np.random.default_rng(seed)
    ids = np.array(tax.ids, dtype=np.int64)
    ix = {t: i for i, t in enumerate(tax.ids)}
    par = np.array([ix[tax.parent[t]] for t in tax.ids], dtype=np.int64)
    strains = np.array([ix[t] for t in tax.ids if tax.rank[t] == "strain"], dtype=np.int64)
    anc = np.zeros((len(ids), 8), dtype=np.int64)          # anc[i, d] = the ancestor d + 1 steps up (the root repeats)
    cur = np.arange(len(ids))
    for d in range(8):
        cur = par[cur]
        anc[:, d] = cur
    genus_of_strain = anc[strains, 1]
    order = np.argsort(genus_of_strain, kind="stable")
    by_genus = strains[order]                                # strains grouped by genus: a genus' 12 strains are neighbours
    pos_in = np.empty(len(ids), dtype=np.int64)
    pos_in[by_genus] = np.arange(by_genus.size)
    want = np.where(rng.random(n_tables) < 0.35, rng.integers(17, 65, n_tables), rng.integers(1, 17, n_tables))
    tab = np.repeat(np.arange(n_tables), want)
    n_ent = tab.size
    base = strains[rng.integers(0, strains.size, n_tables)][tab]
    kind = rng.random(n_ent)
    lvl = rng.integers(0, 7, n_ent)
    rel = by_genus[(pos_in[base] // 12) * 12 + rng.integers(0, 12, n_ent)]
    pick = np.where(kind < 0.45, anc[base, lvl], np.where(kind < 0.55, base, np.where(kind < 0.85, rel, rng.integers(0, len(ids), n_ent))))
    if 374840 in ix:
        pick = np.where(rng.random(n_ent) < 0.002, ix[374840], pick)
    key = np.unique(tab * (1 << 20) + pick)                  # distinct taxids per table
    tab, pick = key >> 20, key & ((1 << 20) - 1)
    shuffle = np.lexsort((rng.random(tab.size), tab))        # registration order is arbitrary
    tab, pick = tab[shuffle], pick[shuffle]
    nT = np.bincount(tab, minlength=n_tables)
    assert nT.min() >= 1 and nT.max() <= 64
    off = np.zeros(n_tables + 1, dtype=np.uint64)
    np.cumsum(nT, out=off[1:])
    cand = rng.integers(1, 1000, n_tables)
    cand = np.where(rng.random(n_tables) < 0.01, rng.integers(1000, 1400, n_tables), cand)     # beyond k4_wave's 999: declined
    levels = rng.integers(1, 6, n_tables)                    # distinct count values of a table
    cnt = (cand[tab] * (1 + rng.integers(0, levels[tab])) // (1 + levels[tab])).astype(np.int64)
    cnt = np.where(rng.random(tab.size) < 0.1, rng.integers(0, cand[tab] + 1), cnt)
    # half of the tables the way reads make them: an ancestor counts at least what its descendants do (the lineage closure adds a
    # position to every ancestor of its ids), so counts fall with depth and relatives of one depth tie
    depth = np.array([tax.depth[t] for t in tax.ids], dtype=np.int64)
    mono = (rng.random(n_tables) < 0.5)[tab]
    cnt_m = np.minimum(cand[tab], cand[tab] * (9 - np.minimum(depth[pick], 8)) // 10 + rng.integers(0, 2, tab.size))
    cnt = np.where(mono, cnt_m, cnt)
    return ids[pick].astype(np.uint32), cnt.astype(np.uint32), off, cand.astype(np.uint32)


def test_k4_wave_equals_the_general_decision_path_on_a_million_tables(tmp_path):
    """The decision step has two forms: k4_part1 / k4_part2 -- a statement-by-statement restatement, pinned to the records the
    reference printed (test above) -- and k4_wave, the wave-parallel form every read of the benchmark takes (partition steps
    of std::sort as ballots, the insertion sort as a rank count, lineage and competitor scan a candidate per lane).  Here both
    run on the same 1.2 M random (taxid, count) tables through lmat_debug_decide_counts: wherever k4_wave takes a table, its
    record equals the general path's bit for bit -- status, match type, call, score, both statistics --; what it declines
    (more than 999 candidate k-mers, the heapsort turn, over-long lineages) comes back marked, never wrong."""
    from lmat_amd import Engine, Params, synth
    tax = synth.make_taxonomy((3, 4, 4, 4, 4, 3), specials=True)
    p = synth.write_aux_files(str(tmp_path), tax)
    eng = Engine(0, Params.run_rl())
    eng.load_taxonomy(p["tree"], p["depth"], p["rank"], p["idmap"])
    fields = ("status", "match_type", "cand_kmer_cnt", "call_tid")
    total = taken = declined_big = 0
    kinds = np.zeros(8, dtype=np.int64)
    for part in range(4):
        tids, cnt, off, cand = _random_count_tables(tax, 300_000, 100 + part)
        gen = eng.debug_decide_counts(tids, cnt, off, cand, on_the_wave=False)
        wav = eng.debug_decide_counts(tids, cnt, off, cand, on_the_wave=True)
        assert (gen["status"] != 255).all()
        took = wav["status"] != 255
        for f in fields:
            assert (wav[f][took] == gen[f][took]).all(), f
        for f in ("call_score", "log_avg", "stdev"):
            assert (wav[f][took].view(np.uint32) == gen[f][took].view(np.uint32)).all(), f
        nT = np.diff(off.astype(np.int64))
        assert not took[cand > 999].any()                       # outside its preconditions: declined, not guessed
        assert took[(cand <= 999) & (nT <= 16)].mean() > 0.97      # ... and inside them nearly everything is taken
        assert took[(cand <= 999) & (nT > 16)].mean() > 0.5
        total += cand.size
        taken += int(took.sum())
        declined_big += int((~took & (cand <= 999)).sum())
        kinds += np.bincount(gen["match_type"][took], minlength=8)[:8]
    assert total == 1_200_000 and taken > 0.9 * total
    assert (kinds[:2] > 1000).all(), kinds     # direct and multi matches both occur among the compared tables (a partial match needs max_val below
                                               # the root's own score, which the max over the entries up to AND including the root never is, :386-404)
    eng.close()


def test_small_integer_division_is_the_ieee_division_on_its_whole_domain():
    """k4_wave scores a taxid count / candidate k-mers (read_label.cpp:821) with a reciprocal, a product and two fused corrections
    instead of the 11-instruction IEEE sequence; both operands are integers below 1024 there.  Every pair of that domain on the
    device: the same bits as the IEEE quotient."""
    from lmat_amd import Engine, Params
    eng = Engine(0, Params.run_rl())
    bad, tried, bad_drawn, drawn = eng.div_check()
    assert tried == 1024 * 1023 and bad == 0
    assert drawn == 1 << 26 and bad_drawn == 0   # the two averages of the statistics: any float >= 0 over 1 .. 64 (proved in kernels.hip; drawn here)
    eng.close()
