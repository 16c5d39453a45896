#!/usr/bin/env python3
"""Generates the reference-pinned fixtures under tests/golden/ (run in the build container):

  ds/               small synthetic dataset (lmat_amd.synth, seeds 1001/2002/3003)
  ref_lookup.txt    per k-mer: taxidCount + taxid() sequence from the REFERENCE's SortedDb::add_data +
                    TaxNodeStat::begin/next (compiled from /root/reference by oracle/Makefile -> oracle/_ref/ref_lookup)
  ref_paths.txt     TaxTree::getPathToRoot for every node of ds/tax.dat, from the reference's TaxTree
  ref_kencode.txt   kencode_c::kencode for 20-mer strings, from the reference's include/kencode.hpp
  ref_tidchecks.txt isHuman / isPhiX truth table from include/tid_checks.hpp
  ref_rkmer[_permissive].txt  per read of ds/reads.fa: valid k-mers, GC bin, marked positions, and the registered
                    taxids in registration order with their position counts, from the REFERENCE's
                    retrieve_kmer_labels (src/rkmer.hpp, compiled in place -> oracle/_ref/ref_rkmer)
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
# ---- make_db_table options: pruning (-g 2 -m rank map), human feed (-j), adaptor feed (-u)
def decode(v, k=20):
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


tax = synth.make_taxonomy((2, 2, 2, 2, 3, 3), True)
with open(os.path.join(ds, "numeric_ranks.txt"), "w") as f:
    for t in tax.ids:
        f.write(f"{t} {tax.depth[t]}\n")
dbk, lists = synth.build_kmer_table(tax, synth.make_genomes(tax, 300, 2002), 20)
rng2 = np.random.default_rng(11)
multi = [int(k) for k, l in zip(dbk.tolist(), lists) if len(l) > 1]
single = [int(k) for k, l in zip(dbk.tolist(), lists) if len(l) == 1]
human = sorted(set(rng2.choice(multi, 150, replace=False).tolist() + rng2.choice(single, 150, replace=False).tolist()
                   + rng2.integers(0, 1 << 40, size=300, dtype=np.uint64).tolist()))
human = [h for h in human if h < int(dbk.max())]  # the reference only merges human k-mers below the last DB k-mer
with open(os.path.join(ds, "human_kmers.txt"), "w") as f:
    f.write("\n".join(decode(h) for h in human) + "\n")
adapt = sorted(set(rng2.choice(multi, 20, replace=False).tolist() + human[5:25:2]))
with open(os.path.join(ds, "adaptor_kmers.txt"), "w") as f:
    f.write("\n".join(decode(a) for a in adapt) + "\n")
q = np.unique(np.concatenate([kms, np.array(human, dtype=np.uint64), np.array(adapt, dtype=np.uint64)]))
np.savetxt(kf, q, fmt="%d")
with open(os.path.join(here, "ref_lookup_opts.txt"), "w") as o:
    out = subprocess.run([os.path.join(ref, "ref_lookup"), "lookup", info["db"], info["idmap"], kf, "200000", "2",
                          os.path.join(ds, "numeric_ranks.txt"), os.path.join(ds, "human_kmers.txt"),
                          os.path.join(ds, "adaptor_kmers.txt")], capture_output=True, text=True, check=True).stdout
    o.write("".join(l + "\n" for l in out.splitlines() if l and l[0].isdigit() and len(l.split()) >= 2 and l.split()[1].isdigit()))
os.remove(kf)
# ---- run-time pruning through the reference's TaxNodeStat::begin (read_label -g 2 [-m ranks])
np.savetxt(kf, kms, fmt="%d")
for tag, env in (("ranks", {"REF_RT_CUT": "2", "REF_RT_RANKS": os.path.join(ds, "numeric_ranks.txt")}), ("noranks", {"REF_RT_CUT": "2"})):
    with open(os.path.join(here, f"ref_lookup_rt_{tag}.txt"), "w") as o:
        out = subprocess.run([os.path.join(ref, "ref_lookup"), "lookup", info["db"], info["idmap"], kf, "200000"],
                             capture_output=True, text=True, check=True, env=dict(os.environ, **env)).stdout
        o.write("".join(l + "\n" for l in out.splitlines() if l and l[0].isdigit() and len(l.split()) >= 2 and l.split()[1].isdigit()))
os.remove(kf)
# ---- the reference's own retrieve_kmer_labels (src/rkmer.hpp) over the fixture reads, default and permissive
for name, perm in (("ref_rkmer.txt", "0"), ("ref_rkmer_permissive.txt", "1")):
    out = subprocess.run([os.path.join(ref, "ref_rkmer"), info["db"], info["idmap"], info["tree"], info["depth"], info["rank"], info["fasta"],
                          "20", perm], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(here, name), "w") as o:
        o.write("".join(l + "\n" for l in out.splitlines() if l.startswith("R ")))
# ---- run-time pruning (-h 2 [-r ranks]) through the same function
for name, extra in (("ref_rkmer_pruned.txt", ["2", os.path.join(ds, "numeric_ranks.txt")]), ("ref_rkmer_pruned_noranks.txt", ["2"])):
    out = subprocess.run([os.path.join(ref, "ref_rkmer"), info["db"], info["idmap"], info["tree"], info["depth"], info["rank"], info["fasta"],
                          "20", "0"] + extra, capture_output=True, text=True, check=True).stdout
    with open(os.path.join(here, name), "w") as o:
        o.write("".join(l + "\n" for l in out.splitlines() if l.startswith("R ")))
# ---- a second, differently shaped dataset (4 strains per species, more sharing, mixed read lengths) for the same function
ds2 = os.path.join(here, "ds2")
info2 = synth.generate_dataset(ds2, (3, 2, 2, 2, 2, 4), 220, 300, L=(60, 100, 150, 250), seeds=(1001, 2102, 3103), frac_short=0.03,
                               lower_frac=0.1, frac_n=0.05, frac_lowc=0.03)
os.remove(info2["fastq"])
for name, perm in (("ref_rkmer_ds2.txt", "0"), ("ref_rkmer_ds2_permissive.txt", "1")):
    out = subprocess.run([os.path.join(ref, "ref_rkmer"), info2["db"], info2["idmap"], info2["tree"], info2["depth"], info2["rank"],
                          info2["fasta"], "20", perm], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(here, name), "w") as o:
        o.write("".join(l + "\n" for l in out.splitlines() if l.startswith("R ")))
# ---- an 18-mer database through the reference's 18-mer index configuration (IDX_CONFIG=1827, SortedDb.hpp:103-109)
ds3 = os.path.join(here, "ds3")
info3 = synth.generate_dataset(ds3, (2, 3, 2, 2, 2, 2), 260, 300, L=(50, 100, 150), k=18, seeds=(1001, 2203, 3203), frac_short=0.03,
                               lower_frac=0.05, frac_n=0.03, frac_lowc=0.03)
os.remove(info3["fastq"])
for name, perm in (("ref_rkmer_ds3.txt", "0"), ("ref_rkmer_ds3_permissive.txt", "1")):
    out = subprocess.run([os.path.join(ref, "ref_rkmer18"), info3["db"], info3["idmap"], info3["tree"], info3["depth"], info3["rank"],
                          info3["fasta"], "18", perm], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(here, name), "w") as o:
        o.write("".join(l + "\n" for l in out.splitlines() if l.startswith("R ")))
print("option lookups:", q.size)
print("k-mers looked up:", kms.size, "db k-mers:", info["n_kmers"], "reads:", len(reads))
