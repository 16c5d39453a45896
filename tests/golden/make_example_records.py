#!/usr/bin/env python3
"""Builds tests/golden/example_records.json from the reference's own example runs (example/example.tgz: two runs of an
LLNL build of read_label over the same 1000 reads, against kML+Human.v4-14.20.g10 and kML.v4-14.20.g10, with null
models and -p; doc/lmat-doc.txt:303-346).

DATA only: per record every field the reference printed except the read -- header, the statistics column
(log_avg, stdev, distinct k-mers; text as printed), the candidate list (taxid, score text) in printed order, and the call
(taxid, score text, match type) -- plus, from the first run's .fastsummary / .fastsummary.lineage, the names of the called
taxids and the ranked lineages the reference's own post-processing printed for them.  Run in the build container."""
import glob
import json
import os
import sys
import tarfile
import tempfile

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/example/example.tgz"
here = os.path.dirname(os.path.abspath(__file__))
out = {"runs": {}}
with tempfile.TemporaryDirectory() as td:
    tarfile.open(ref).extractall(td)
    for run in ("kML+Human.v4-14.20.g10", "kML.v4-14.20.g10"):
        pre = "simple_list.1000.fna.%s.db.lo.rl_output" % run
        recs = []
        for f in sorted(glob.glob(os.path.join(td, pre + "[0-9].out"))):
            for line in open(f):
                c = line.rstrip("\n").split("\t")
                assert len(c) in (4, 5), line
                cand = c[3].split() if len(c) == 5 else []
                recs.append({"hdr": c[0], "stats": c[2].split(), "cands": [[cand[i], cand[i + 1]] for i in range(0, len(cand), 2)],
                             "call": c[-1].split()})
        out["runs"][run] = recs
    pre = "simple_list.1000.fna.kML+Human.v4-14.20.g10.db.lo.rl_output.0.30.fastsummary"
    out["names"] = {l.split("\t")[2]: l.rstrip("\n").split("\t")[3] for l in open(os.path.join(td, pre))}
    out["lineages"] = [l.rstrip("\n").split("\t")[1:] for l in open(os.path.join(td, pre + ".lineage"))]
json.dump(out, open(os.path.join(here, "example_records.json"), "w"), separators=(",", ":"))
print({k: len(v) for k, v in out["runs"].items()}, len(out["names"]), "names", len(out["lineages"]), "lineages")
