import numpy as np, sys
sys.path.insert(0, '.')
from lmat_amd import Engine, Params
eng = Engine(0, Params.run_rl())   # -p: n_cand = registered taxids with score >= 0 = nT
eng.synth_taxonomy((3, 4, 4, 4, 4, 3))
tb = int(__import__("os").environ.get("NT_DB_GB", "8")) << 30
G = int(0.8 * (tb / 8) / (768 * (1.0 + 3 * (1 - 0.99 ** 20))))
eng.synth_db(G, k=20, seed=2002, table_bytes=tb)
reads = eng.synth_reads(200000, (150,), seed=3003)
res, cands = eng.classify(reads, cand_cap=200000 * 40)
n = res["n_cand"][res["status"] == 0]
h = np.bincount(n, minlength=70)
c = np.cumsum(h) / h.sum()
for t in (8, 12, 16, 20, 24, 28, 32, 40, 48, 64):
    print(t, round(float(c[t]), 4))
print("mean", n.mean(), "max", n.max())
