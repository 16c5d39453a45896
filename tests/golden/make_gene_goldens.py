#!/usr/bin/env python3
"""Generates the gene_label fixtures under tests/golden/gene/ (run in the build container):

  gene.bin            a small gene database in the tax_histo record format (k = 20): k-mer -> list of 32-bit gene ids,
                      some above 2^24 and 2^31; gene families share segments, so many k-mers carry several ids
  ref_lookup_gene.txt per k-mer: count + id sequence from the REFERENCE's SortedDb<uint32_t>::add_data (no 32->16 map) +
                      TaxNodeStat<uint32_t>::begin/next -- the TID_SIZE=32 build gene_label links -- compiled from
                      /root/reference by oracle/Makefile -> oracle/_ref/ref_lookup32
  rl0.out, rl1.out    read_label-format records (header, read, statistics, candidates, call) over reads cut from the genes,
                      random reads, short reads: gene_label's input (-l list of these)
  genes.tbl.gz        the gene annotation table gene_label joins its summaries against (-g)
Only inputs and expected outputs are stored; no reference source travels."""
import gzip
import os
import subprocess
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "oracle"))
from lmat_amd import synth  # noqa: E402

subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "all"], stdout=subprocess.DEVNULL)
out = os.path.join(here, "gene")
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(515)
k = 20
n_fam, per_fam, glen = 14, 4, 700
genes, gid = {}, []
for f in range(n_fam):
    core = rng.integers(0, 4, glen, dtype=np.uint8)
    for m in range(per_fam):
        g = core.copy()
        lo = int(rng.integers(0, glen - 200))
        g[lo:lo + 200] = rng.integers(0, 4, 200, dtype=np.uint8)        # a member-specific stretch
        mut = rng.random(glen) < 0.01
        g[mut] = (g[mut] + rng.integers(1, 4, int(mut.sum()))) & 3
        ident = [100000 + 7919 * (f * per_fam + m), 20000000 + 31 * (f * per_fam + m), 3000000000 + 977 * (f * per_fam + m)][(f + m) % 3]
        genes[ident] = g
        gid.append(ident)
table = {}
for ident in gid:
    for km in np.unique(synth.kmers_of(genes[ident], k)).tolist():
        table.setdefault(km, []).append(ident)
kmers = np.array(sorted(table), dtype=np.uint64)
lists = [table[int(x)] for x in kmers]
db = os.path.join(out, "gene.bin")
synth.write_taxhisto(db, kmers, lists, k)
# golden lookups from the reference's 32-bit build
probe = np.unique(np.concatenate([kmers[::3], rng.integers(0, 1 << 40, size=400, dtype=np.uint64)]))
kf = os.path.join(out, "_kmers.tmp")
np.savetxt(kf, probe, fmt="%d")
txt = subprocess.run([os.path.join(root, "oracle", "_ref", "ref_lookup32"), db, kf, "200000"], capture_output=True, text=True, check=True).stdout
open(os.path.join(out, "ref_lookup_gene.txt"), "w").write("".join(l + "\n" for l in txt.splitlines() if l and l[0].isdigit()))
os.remove(kf)
# read_label-format input
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
recs = []
for i in range(900):
    u = rng.random()
    if u < 0.75:
        g = genes[gid[int(rng.integers(0, len(gid)))]]
        L = int(rng.choice([60, 100, 150, 250]))
        o = int(rng.integers(0, glen - L))
        s = g[o:o + L].copy()
        e = rng.random(L) < 0.01
        s[e] = (s[e] + 1) & 3
        if rng.random() < 0.5:
            s = (3 - s)[::-1]
        seq = letters[s].tobytes().decode()
        if rng.random() < 0.05:
            p = int(rng.integers(0, L))
            seq = seq[:p] + "N" + seq[p + 1:]
    elif u < 0.9:
        seq = letters[rng.integers(0, 4, 150)].tobytes().decode()
    else:
        seq = letters[rng.integers(0, 4, int(rng.integers(5, 19)))].tobytes().decode()
    # the columns gene_label parses: statistics (third field -1 = skip), candidates, call "<taxid> <score> <type>"
    if len(seq) < k:
        rec = f"r{i}\t{seq}\t-1 -1 -1\t-1 -1\t{len(seq)} {k} ReadTooShort"
    elif rng.random() < 0.1:
        rec = f"r{i}\t{seq}\t-1 -1 {len(seq) - k + 1}\t-1 -1\t{len(seq)} {k} NoDbHits"
    else:
        tid = int(rng.choice([562, 5476, 9606, 1280]))
        sc = float(rng.choice([0.25, 0.5, 0.731, 1.0, 1.23277]))
        kind = str(rng.choice(["DirectMatch", "MultiMatch"]))
        rec = f"r{i}\t{seq}\t0.5 0.1 {len(seq) - k + 1}\t {tid} {sc:g}\t{tid} {sc:g} {kind}"
    recs.append(rec)
open(os.path.join(out, "rl0.out"), "w").write("\n".join(recs[:500]) + "\n")
open(os.path.join(out, "rl1.out"), "w").write("\n".join(recs[500:]) + "\n")
with gzip.open(os.path.join(out, "genes.tbl.gz"), "wt") as g:
    for n, ident in enumerate(gid):
        g.write(f"{562 + n} {ident} gene{n} family{n // per_fam} some annotation text\n")
print(len(kmers), "k-mers,", len(gid), "genes,", len(recs), "records")
