"""N>1 path on CPU: two gloo ranks each label their shard (CPU oracle as the per-rank worker), the dense tallies are
all-reduced with the same helper bench.py uses on RCCL, and the merged result equals the single-process run."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
from lmat_amd.shard import shard_range, allreduce_tallies
import oracle_py
ds = json.load(open(sys.argv[2]))
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
reads = ds["reads"]
lo, hi = shard_range(len(reads), rank, world)
o = oracle_py.Oracle(ds["tree"], ds["depth"], ds["rank"], ds["idmap"]); o.add_taxhisto(ds["db"]); o.set_options()
bs = [r.encode() for r in reads[lo:hi]]
off = np.zeros(len(bs) + 1, dtype=np.uint64); np.cumsum([len(b) for b in bs], out=off[1:])
text, tally, nm = o.classify(np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8), off, 20, first_index=lo)
ids = sorted(ds["ids"])
index = {t: i for i, t in enumerate(ids)}
cnt = torch.zeros(len(ids), dtype=torch.int64); sc = torch.zeros(len(ids), dtype=torch.float64)
for t, (c, s) in tally.items():
    cnt[index[t]] = c; sc[index[t]] = s
nmt = torch.tensor(nm, dtype=torch.int64)
allreduce_tallies(cnt, sc, nmt, dist)
open(sys.argv[3] + f".{rank}.out", "w").write(text)
if rank == 0:
    json.dump({"count": {str(ids[i]): int(c) for i, c in enumerate(cnt.tolist()) if c}, "nomatch": nmt.tolist(),
               "score": {str(ids[i]): float(s) for i, s in enumerate(sc.tolist()) if s}}, open(sys.argv[3] + ".json", "w"))
dist.destroy_process_group()
'''


def test_shard_ranges_partition():
    from lmat_amd.shard import shard_range
    for n in (0, 1, 7, 1000):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_two_rank_gloo_merge_equals_single_process(small_dataset, oracle_small, tmp_path):
    import json
    ds = dict(small_dataset)
    ids = set()
    for line in open(ds["idmap"]):
        ids.add(int(line.split()[0]))
    ids.add(0)
    ds["ids"] = sorted(ids)
    cfg = tmp_path / "ds.json"
    cfg.write_text(json.dumps(ds))
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket() as sk:  # a free port: concurrent runs on one host must not collide
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(w), ROOT, str(cfg), out],
                          env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
    reads = ds["reads"]
    bs = [r.encode() for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    text, tally, nm = oracle_small.classify(np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8), off, 20)
    merged = json.load(open(out + ".json"))
    assert merged["nomatch"] == nm
    assert {int(t): c for t, c in merged["count"].items()} == {t: c for t, (c, s) in tally.items()}
    for t, s in merged["score"].items():
        assert abs(s - tally[int(t)][1]) < 1e-3
    # per-read records: the two shards concatenated in rank order are the single-process output
    assert open(out + ".0.out").read() + open(out + ".1.out").read() == text
