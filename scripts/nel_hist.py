#!/usr/bin/env python3
"""Histogram of kept-list elements per read (sum of list lengths over a read's distinct lists) on the bench workload:
which capacity class a read needs.  Run with LMAT_STOP_AFTER=7 (debug stop after the list headers are read)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LMAT_STOP_AFTER"] = "7"
from lmat_amd import Engine, Params
import bench

db_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
k = 20
eng = Engine(0, Params.run_rl(prn_all=0))
eng.synth_taxonomy(bench.BRANCHING)
table_bytes = int(db_gb * (1 << 30)) // 64 * 64
n_species, S = 768, 3
pm = 1.0 - 0.99 ** k
G = int(0.8 * (table_bytes / 8) / (n_species * (1.0 + S * pm)))
eng.synth_db(G, k=k, seed=2002, table_bytes=table_bytes)
reads = eng.synth_reads(n, (150,), seed=3003)
res, _ = eng.classify(reads, want_cands=False)
sel = res["status"] == 250
nel = res["cand_kmer_cnt"][sel].astype(np.int64)
print("reads", n, "with hits", int(sel.sum()))
for t in (16, 32, 64, 96, 128, 192, 256, 320, 384, 448, 512, 768, 1024):
    print(f"nel <= {t}: {(nel <= t).mean():.4f}")
print("max nel", nel.max(), "mean", nel.mean())
