#!/usr/bin/env python3
"""Histogram of candidate-table sizes (registered taxids per read) on the bench workload: which K4 kernel a read lands in."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmat_amd import Engine, Params
import bench

db_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
k = 20
eng = Engine(0, Params.run_rl(prn_all=1))
eng.synth_taxonomy(bench.BRANCHING)
table_bytes = int(db_gb * (1 << 30)) // 64 * 64
n_species, S = 768, 3
pm = 1.0 - 0.99 ** k
G = int(0.8 * (table_bytes / 8) / (n_species * (1.0 + S * pm)))
eng.synth_db(G, k=k, seed=2002, table_bytes=table_bytes)
reads = eng.synth_reads(n, (150,), seed=3003)
res, cands = eng.classify(reads, cand_cap=64 * n)
nc = res["n_cand"]
st = res["status"]
print("status counts:", dict(zip(*np.unique(st, return_counts=True))))
h = np.bincount(np.minimum(nc, 80), minlength=81)
cum = np.cumsum(h) / n
for t in (8, 12, 16, 20, 24, 28, 32, 40, 48, 64):
    print(f"n_cand <= {t}: {cum[min(t, 80)]:.4f}")
print("max n_cand", nc.max())
