#!/usr/bin/env python3
"""Build-container only: how fast is the CPU restatement (oracle/) next to the REFERENCE's own code on the same input?

read_label.cpp cannot be built in this image (it needs generated headers, gzstream and perm-je: DESIGN.md section 5), so
there is no whole-path reference timing.  What does build from the reference's own files is src/rkmer.hpp's
retrieve_kmer_labels -- k-mer extraction, per-read dedupe, SortedDb lookup + TaxNodeStat conversion, taxid filtering, depth
sort, leaf-most filter, representative strain, lineage closure: everything of proc_line before the scoring, and where a CPU
run spends its time -- as oracle/_ref/ref_rkmer.  This script times that per-read loop (single thread, database resident,
loading outside the window) and the oracle's restatement of the same function on the same reads, and prints the ratio that
turns a `cpu_baseline` of kind "port" into reference-equivalent reads/s.  TIMING ONLY: it pins nothing and nothing of it ships."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
G = os.path.join(ROOT, "tests", "golden")


def one(ds, exe, k, passes):
    d = os.path.join(G, ds)
    f = {n: os.path.join(d, v) for n, v in dict(db="th.bin", idmap="map32to16.txt", tree="tax.dat", depth="depth.dat", rank="rank.txt", fasta="reads.fa").items()}
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe), f["db"], f["idmap"], f["tree"], f["depth"], f["rank"], f["fasta"], str(k)],
                       env=dict(os.environ, LMAT_REF_TIME=str(passes)), capture_output=True, text=True, check=True)
    ref = float(r.stderr.split("reads_per_s")[1])
    import oracle_py
    orc = oracle_py.Oracle(f["tree"], f["depth"], f["rank"], f["idmap"])
    orc.add_taxhisto(f["db"])
    reads = [l.rstrip("\n") for l in open(f["fasta"]) if not l.startswith(">")]
    bs = [x.encode() for x in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    blob = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8)
    orc.rkmer_trace(blob, off, k, False)
    t0 = time.perf_counter()
    for _ in range(passes):
        orc.rkmer_trace(blob, off, k, False)
    port = len(reads) * passes / (time.perf_counter() - t0)
    orc.close()
    return {"dataset": ds, "k": k, "reads": len(reads), "passes": passes, "reference_reads_per_s": ref, "port_reads_per_s": port, "port_over_reference": port / ref}


if __name__ == "__main__":
    rows = [one("ds", "ref_rkmer", 20, 200), one("ds2", "ref_rkmer", 20, 100), one("ds3", "ref_rkmer18", 18, 200)]
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
    print(json.dumps({"cpu": cpu, "threads": 1, "rows": rows, "geomean_port_over_reference": float(np.exp(np.mean([np.log(r["port_over_reference"]) for r in rows])))}, indent=1))
