#!/usr/bin/env python3
"""Builds tests/golden/example_kmer_counts.tsv and example_summary.json from the reference's own example run
(example/example.tgz: inputs + the outputs an LLNL build of read_label produced, doc/lmat-doc.txt:303-346).

The fixture is DATA only: per read the header, the sequence and the k-mer count column the reference printed
(distinct valid 20-mers for classified reads, valid 20-mers for NoDbHits / ReadTooShort rows), plus the
per-record call (taxid, score, type) and the .fastsummary / .nomatchsum the reference derived from them.
Run in the build container (needs /root/reference)."""
import glob
import json
import os
import sys
import tarfile
import tempfile

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/example/example.tgz"
here = os.path.dirname(os.path.abspath(__file__))
with tempfile.TemporaryDirectory() as td:
    tarfile.open(ref).extractall(td)
    pre = "simple_list.1000.fna.kML+Human.v4-14.20.g10.db.lo.rl_output"
    rows, calls, stats = [], [], []
    for f in sorted(glob.glob(os.path.join(td, pre + "[0-7].out"))):
        for line in open(f):
            c = line.rstrip("\n").split("\t")
            hdr, read, st, last = c[0], c[1], c[2].split(), c[-1].split()
            kind = last[-1]
            if kind == "ReadTooShort":
                what, exp = "valid", int(last[0])
            elif kind == "NoDbHits":
                what, exp = "valid", int(st[2])
            else:
                what, exp = "cand", int(st[2])
                calls.append([int(last[0]), last[1], kind])
                cand = c[3].split() if len(c) == 5 else []
                if st[0] != "-1" and cand and cand[0] != "-1":
                    # the statistics columns next to the candidate scores the same record prints (text as printed)
                    stats.append([st[0], st[1], cand[1::2]])
            rows.append((hdr, read, kind, what, exp))
    with open(os.path.join(here, "example_kmer_counts.tsv"), "w") as o:
        for r in rows:
            o.write("\t".join(str(x) for x in r) + "\n")
    fs = [l.rstrip("\n").split("\t")[:3] for l in open(os.path.join(td, pre + ".0.30.fastsummary"))]
    nm = dict(l.split() for l in open(os.path.join(td, pre + ".0.30.nomatchsum")))
    json.dump({"calls": calls, "fastsummary": fs, "nomatchsum": nm, "min_score": 0.0},
              open(os.path.join(here, "example_summary.json"), "w"))
    json.dump(stats, open(os.path.join(here, "example_score_stats.json"), "w"))
    print(len(rows), "reads,", len(calls), "calls")
