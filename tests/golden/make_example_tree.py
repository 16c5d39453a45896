#!/usr/bin/env python3
"""Builds tests/golden/example_tree.json from the reference's own example run (example/example.tgz):

  * `...rl_output.0.30.fastsummary.summ` -- content_summ's report: the taxonomy tree of everything the run called, one node per
    line, indentation = parent (src/content_summ.cpp:452-486): name, TaxID, Reads, WReads.  34 nodes with their true
    parents: Candida albicans' whole lineage with three strains, Pseudomonas putida down to strain ND6, a herpesvirus, the
    synthetic-construct branch.
  * the per-rank k-mer coverage tables content_summ wrote beside it (`.summ.<rank>_kmer_cov`), the log, and the roll-ups
    bin/run_rl.sh makes from the .fastsummary (`.lineage`, `.species`, `.genus`, `.ordered.*`): kept verbatim as data.

DATA only (report files of the reference's run, no source).  Run in the build container."""
import json
import os
import sys
import tarfile
import tempfile

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/example/example.tgz"
here = os.path.dirname(os.path.abspath(__file__))
pre = "simple_list.1000.fna.kML+Human.v4-14.20.g10.db.lo.rl_output.0.30.fastsummary"
out = {"nodes": [], "files": {}}
with tempfile.TemporaryDirectory() as td:
    tarfile.open(ref).extractall(td)
    lines = open(os.path.join(td, pre + ".summ")).read().split("\n")
    assert lines[0] == "Name\tTaxID\tReads\tWReads"
    stack = []  # taxid at each indentation level
    for ln in lines[1:]:
        if not ln:
            continue
        depth = len(ln) - len(ln.lstrip("\t"))
        name, tid, reads, wreads = ln.lstrip("\t").split("\t")
        stack = stack[:depth]
        out["nodes"].append({"tid": int(tid), "name": name, "parent": stack[-1] if stack else int(tid), "depth": depth,
                             "reads": int(reads), "wreads": wreads})
        stack.append(int(tid))
    for f in sorted(os.listdir(td)):
        if f.startswith(pre) and f != pre + ".html":
            out["files"][f[len(pre):]] = open(os.path.join(td, f)).read()
json.dump(out, open(os.path.join(here, "example_tree.json"), "w"), indent=0, separators=(",", ":"))
print(len(out["nodes"]), "nodes;", len(out["files"]), "report files")
