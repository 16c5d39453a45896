#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 CSVs) into tracked files under profiles/."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{src}/kt/kt_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
shutil.copy(f"{src}/kt_bench.json", f"profiles/{tag}_bench_under_rocprof.json")
pmc = {}
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{src}/{sub}/{sub}_counter_collection.csv")):
        if r["Counter_Name"] == ctr:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    pmc[ctr] = {k: {"dispatches": len(v), "mean_KiB_per_dispatch": sum(v) / len(v)} for k, v in agg.items()}
bench = json.loads(open(f"{src}/kt_bench.json").read().strip().splitlines()[-1])
cls = max((k for k in pmc["FETCH_SIZE"] if "classify_kernel" in k), key=lambda k: pmc["FETCH_SIZE"][k]["mean_KiB_per_dispatch"])  # the fast class
gat = [k for k in pmc["FETCH_SIZE"] if "gather_bench" in k]
fetch = pmc["FETCH_SIZE"][cls]["mean_KiB_per_dispatch"] * 1024
write = pmc["WRITE_SIZE"][cls]["mean_KiB_per_dispatch"] * 1024
out = {
    "tag": tag,
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py " + " ".join(sys.argv[2:]),
    "kernel": cls,
    "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
    "reads_per_launch": bench["roofline"]["reads_per_launch"],
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_read"] * bench["roofline"]["reads_per_launch"],
    "calibration": None, "pmc": pmc,
}
if gat:
    known = (((1 << 28) // 4096 + 143) // 144 * 144) * 4096 * 64  # lmat_gather_bench rounds the probes per wave up to 144
    meas = pmc["FETCH_SIZE"][gat[0]]["mean_KiB_per_dispatch"] * 1024
    out["calibration"] = {"kernel": gat[0], "known_bytes": known, "FETCH_SIZE_bytes": meas, "ratio": meas / known,
                          "note": "random 64-B bucket gather, 4 lanes x 16 B (8 x 8 B up to r01e): FETCH_SIZE x 1024 equals the byte count "
                                  "(the x2 correction of the guide applies to 16-B/lane streaming reads, not to this shape)"}
json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
json.dump({"hbm_bytes_per_launch": fetch + write, "from": f"profiles/{tag}_traffic.json", "reads_per_launch": bench["roofline"]["reads_per_launch"],
           "db_gib": bench["config"]["db_gib"], "read_len": bench["config"]["read_len"], "kernel_avg_ms": bench["roofline"]["kernel_avg_ms"]},
          open("profiles/traffic_latest.json", "w"))
print(json.dumps({k: out[k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "algorithmic_bytes_per_launch", "calibration")}, indent=1))
