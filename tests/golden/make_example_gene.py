#!/usr/bin/env python3
"""Builds tests/golden/example_gene.tar.gz from the reference's own example run (example/example.tgz): the eight read_label
output files gene_label was given (in the order of rl_output.flst), the eight gene_label output files, the two gene summaries,
and the gene annotation table as far as the run shows it (the summary lines carry the table's lines verbatim; the second
summary is in table order).
Data only; run in the build container:  python tests/golden/make_example_gene.py"""
import io, os, tarfile

REF = "/root/reference/example/example.tgz"
HERE = os.path.dirname(os.path.abspath(__file__))

src = tarfile.open(REF)
files = {os.path.basename(m.name): src.extractfile(m).read() for m in src.getmembers() if m.isfile()}
flst = files["rl_output.flst"].decode().split()
GL = "rl_output.flst.allgenes.7-14.20.db.gl_output"
out = {}
for i, fn in enumerate(flst):
    out[f"rl{i}.out"] = files[os.path.basename(fn)]
    out[f"gl{i}.out"] = files[f"{GL}{i}.out"]
out["genesummary"] = files[f"{GL}.0.1.20.genesummary"]
out["genesummary_tax"] = files[f"{GL}.0.1.20.genesummary.min_tax_score.0"]
# bin/run_gl.sh:160-161 sorts the first summary by its score column afterwards (sort -k1gr,1gr); the second one is as
# gene_label wrote it, i.e. in table order.  Table lines only the first summary shows go behind those, in any order:
# their place does not show in either file.
table, seen = [], set()
for name in ("genesummary_tax", "genesummary"):
    for line in out[name].decode().splitlines():
        row = line.split("\t", 3)[3]  # avg \t count \t taxid \t <table line>
        if row not in seen:
            seen.add(row)
            table.append(row)
out["genes.tbl"] = ("\n".join(table) + "\n").encode()
out["README"] = (b"From example/example.tgz of the reference: rl<i>.out = i-th file of rl_output.flst, gl<i>.out = gene_label's output for it,\n"
                 b"genesummary[_tax] = <base>.0.1.20.genesummary[.min_tax_score.0] (-x 0.1 -q 20 -b 0), genes.tbl = the table lines the summaries show.\n")
with tarfile.open(os.path.join(HERE, "example_gene.tar.gz"), "w:gz") as t:
    for k in sorted(out):
        ti = tarfile.TarInfo(k)
        ti.size = len(out[k])
        t.addfile(ti, io.BytesIO(out[k]))
print("wrote example_gene.tar.gz:", {k: len(v) for k, v in out.items()})
