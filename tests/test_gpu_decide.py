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
