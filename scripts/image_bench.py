#!/usr/bin/env python3
"""Start-up from a device image (LMATIMG2) at full size: build the synthetic database of `bench.py` (--db-gb, default 64), stream
it out of HBM into a file, stream it back into a fresh context, check that the copy answers like the original (lookups of a
read sample + classification of 200 k reads: identical result records), and report GB/s of both directions next to what building
the same table costs.  The reference opens its database by mmap (src/read_label.cpp:1477-1491); this is the counterpart.
usage: image_bench.py [db_gb] [directory]   -> one JSON line"""
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lmat_amd import Engine, Params  # noqa: E402
import bench  # noqa: E402

db_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 64.0
need = int(db_gb * 1.35 * (1 << 30))
cands = [sys.argv[2]] if len(sys.argv) > 2 else [os.environ.get("TMPDIR", "/tmp"), "/dev/shm"]
where = next((d for d in cands if os.path.isdir(d) and shutil.disk_usage(d).free > need), None)
if where is None:
    sys.exit(f"image_bench: no directory with {need >> 30} GiB free among {cands}")
fn = os.path.join(where, "lmat_bench.img2")
k = 20
table_bytes = int(db_gb * (1 << 30)) // 64 * 64
G = int(0.8 * (table_bytes / 8) / (768 * (1.0 + 3 * (1.0 - 0.99 ** k))))
a = Engine(0, Params.run_rl(prn_all=0))
a.synth_taxonomy(bench.BRANCHING)
t0 = time.perf_counter()
a.synth_db(G, k=k, seed=2002, table_bytes=table_bytes)
t_build = time.perf_counter() - t0
reads = a.synth_reads(200_000, (150,), seed=3003)
res_a, _ = a.classify(reads, want_cands=False)
blob, off = reads.ascii(0, 200_000)
t0 = time.perf_counter()
a.save_device_image(fn)
t_save = time.perf_counter() - t0
size = os.path.getsize(fn)
try:
    b = Engine(0, Params.run_rl(prn_all=0))
    b.synth_taxonomy(bench.BRANCHING)
    t0 = time.perf_counter()
    b.load_image(fn)
    t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    b2 = None
    if os.environ.get("LMAT_IMAGE_BENCH_TWICE", "1") == "1":   # a second load: the file is in the page cache by now at the latest
        b2 = Engine(0, Params.run_rl(prn_all=0)) if db_gb <= 100 else None
        if b2 is not None:
            a.close()
            a = None
            b2.synth_taxonomy(bench.BRANCHING)
            t0 = time.perf_counter()
            b2.load_image(fn)
    t_load2 = time.perf_counter() - t0 if b2 is not None else None
    rb = b.upload_reads((np.append(blob, np.uint8(0)), off))
    res_b, _ = b.classify(rb, want_cands=False)
    same = bool((res_a.view(np.uint8) == res_b.view(np.uint8)).all())
    out = {"db_gib": db_gb, "db_kmers": b.db_size, "image_bytes": size, "image_gib": size / 2**30, "directory": where,
           "build_s": t_build, "save_s": t_save, "save_GBs": size / t_save / 1e9, "load_s": t_load, "load_GBs": size / t_load / 1e9,
           "second_load_s": t_load2, "second_load_GBs": size / t_load2 / 1e9 if t_load2 else None,
           "results_identical_on_200k_reads": same,
           "what": "lmat_db_save_image on the finalized database (HBM -> pinned 64 MiB chunks -> pwrite, 4 threads) and lmat_db_load_image "
                   "(pread -> pinned -> HBM) of the LMATIMG2 device layout; build_s = generating and inserting the same table on the device"}
    print(json.dumps(out))
    assert same
finally:
    os.unlink(fn)
