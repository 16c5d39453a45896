#!/usr/bin/env python3
"""Generates the reference-pinned fixtures under tests/golden/ (run in the build container):

  ds/               small synthetic dataset (lmat_amd.synth, seeds 1001/2002/3003)
  ref_lookup.txt    per k-mer: taxidCount + taxid() sequence from the REFERENCE's SortedDb::add_data +
                    TaxNodeStat::begin/next (compiled from /root/reference by oracle/Makefile -> oracle/_ref/ref_lookup)
  ref_paths.txt     TaxTree::getPathToRoot for every node of ds/tax.dat, from the reference's TaxTree
  ref_kencode.txt   kencode_c::kencode for 20-mer strings, from the reference's include/kencode.hpp
  ref_tidchecks.txt isHuman / isPhiX truth table from include/tid_checks.hpp
Only inputs and expected outputs are stored; no reference source travels."""
import os
import subprocess
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "oracle"))
from lmat_amd import synth  # noqa: E402
import oracle_py  # noqa: E402

subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "all"], stdout=subprocess.DEVNULL)
ref = os.path.join(root, "oracle", "_ref")
ds = os.path.join(here, "ds")
info = synth.generate_dataset(ds, (2, 2, 2, 2, 3, 3), 300, 300, frac_short=0.03, lower_frac=0.05)
reads = [l.rstrip("\n") for l in open(info["fasta"]) if not l.startswith(">")]
orc = oracle_py.Oracle(info["tree"], info["depth"], info["rank"], info["idmap"])
kms = [orc.extract(r.encode(), 20)[0] for r in reads]
kms = np.unique(np.concatenate(kms))
rng = np.random.default_rng(7)
kms = np.unique(np.concatenate([kms, rng.integers(0, 1 << 40, size=500, dtype=np.uint64)]))
kf = os.path.join(here, "_kmers.tmp")
np.savetxt(kf, kms, fmt="%d")
with open(os.path.join(here, "ref_lookup.txt"), "w") as o:
    out = subprocess.run([os.path.join(ref, "ref_lookup"), "lookup", info["db"], info["idmap"], kf, "200000"],
                         capture_output=True, text=True, check=True).stdout
    o.write("".join(l + "\n" for l in out.splitlines() if l and l[0].isdigit()))
os.remove(kf)
with open(os.path.join(here, "ref_paths.txt"), "w") as o:
    o.write(subprocess.run([os.path.join(ref, "ref_lookup"), "paths", info["tree"]], capture_output=True, text=True,
                           check=True).stdout)
with open(os.path.join(here, "ref_tidchecks.txt"), "w") as o:
    o.write(subprocess.run([os.path.join(ref, "ref_lookup"), "tidchecks"], capture_output=True, text=True, check=True).stdout)
# kencode: 20-mers cut from the reads (upper and lower case) and their reverse complements
comp = str.maketrans("ACGTacgt", "TGCAtgca")
strs = []
for r in reads[:80]:
    if len(r) >= 40 and set(r.upper()) <= set("ACGT"):
        for p in (0, 7, len(r) - 20):
            s = r[p:p + 20]
            strs += [s, s.translate(comp)[::-1]]
sf = os.path.join(here, "_strs.tmp")
open(sf, "w").write("\n".join(strs) + "\n")
with open(os.path.join(here, "ref_kencode.txt"), "w") as o:
    o.write(subprocess.run([os.path.join(ref, "ref_kencode"), "20", sf], capture_output=True, text=True, check=True).stdout)
os.remove(sf)
print("k-mers looked up:", kms.size, "db k-mers:", info["n_kmers"], "reads:", len(reads))
