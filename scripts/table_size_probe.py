import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from lmat_amd import Engine, Params
G = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden")
DS = os.path.join(G, "ds")
t0 = time.time()
eng = Engine(0, Params.run_rl())
eng.load_taxonomy(os.path.join(DS, "tax.dat"), os.path.join(DS, "depth.dat"), os.path.join(DS, "rank.txt"), os.path.join(DS, "map32to16.txt"))
print("tax", time.time() - t0, flush=True)
eng.build_db(os.path.join(DS, "th.bin"), k=20, table_bytes=int(float(sys.argv[1]) * 2**30), n_kmers_hint=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
print("built", time.time() - t0, eng.db_size, flush=True)
kms, want = [], []
for line in open(os.path.join(G, "ref_lookup.txt")):
    f = line.split(); kms.append(int(f[0])); want.append([int(x) for x in f[2:]])
counts, tids = eng.lookup(np.array(kms, dtype=np.uint64), stride=32)
bad = sum(1 for i, w in enumerate(want) if counts[i] != len(w) or tids[i, :len(w)].tolist() != w)
print("lookup mismatches", bad, time.time() - t0, flush=True)
eng.close()
