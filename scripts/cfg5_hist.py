#!/usr/bin/env python3
"""Which capacity class do the reads of BASELINE config 5 (75..300 bp, genus-shared blocks) need?  Per read length: distinct taxid
lists per read (LMAT_STOP_AFTER=4), kept-list elements (=7) and registered taxids (-p candidates), with the shares beyond the
fast classes' limits (64 lists / 256 elements / 64 taxids) and beyond the re-run class's.  The distribution follows from the
generator's mutation pattern, not from the table size: an 8.3 GiB table is enough."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmat_amd import Engine, Params
import bench

db_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 8.3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
lens = (75, 100, 125, 150, 200, 250, 300)
k = 20
eng = Engine(0, Params.run_rl(prn_all=1))
eng.synth_taxonomy(bench.BRANCHING)
table_bytes = int(db_gb * (1 << 30)) // 64 * 64
G = int(0.8 * (table_bytes / 8) / (768 * (1.0 + 3 * (1.0 - 0.99 ** k))))
eng.synth_db(G, k=k, seed=2002, table_bytes=table_bytes)
reads = eng.synth_reads(n, lens, seed=3003)
out = {}
for name, stop in (("ndist", "4"), ("nel", "7"), ("nT", None)):
    if stop:
        os.environ["LMAT_STOP_AFTER"] = stop
    else:
        os.environ.pop("LMAT_STOP_AFTER", None)
    res, _ = eng.classify(reads, cand_cap=80 * n)
    if stop:
        sel = res["status"] == 250
        v = res["cand_kmer_cnt"].astype(np.int64)
    else:
        sel = res["status"] == 0
        v = res["n_cand"].astype(np.int64)
    out[name] = (sel, v, res["read_len"])
for name, lims in (("ndist", (32, 64, 128, 256)), ("nel", (128, 256, 512, 1024, 2048, 4096)), ("nT", (16, 32, 64, 128))):
    sel, v, rl = out[name]
    print(f"== {name}: reads with a value {int(sel.sum())} of {n}")
    for L in lens + (0,):
        m = sel & (rl == L) if L else sel
        x = v[m]
        if not x.size:
            continue
        print(f"  len {L or 'all':>4}: mean {x.mean():7.1f} p50 {np.percentile(x, 50):6.0f} p99 {np.percentile(x, 99):6.0f} max {x.max():6d} | " +
              " ".join(f">{t}: {(x > t).mean() * 100:6.3f}%" for t in lims))
reads.free()
eng.close()
