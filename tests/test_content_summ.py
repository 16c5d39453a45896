"""CPU: content_summ (src/content_summ.cpp; bin/run_cs.sh:148) against the reference's own example run.

Inputs as that run had them: the eight read_label .out files (tests/golden/example_gene.tar.gz holds them in the order of
rl_output.flst), the run's .fastsummary, and -k 8,10,12,14,17 -a plasmid,species,genus as run_cs.sh passes them.  The two
runtime inputs the repository does not hold are rebuilt from what the run printed (tests/golden/example_tree.json):
  * the taxonomy tree = the .summ report itself (indentation = parent, names as printed);
  * the rank table = the rank each called taxid carries in the .fastsummary ("species,Candida albicans"; blanks are
    underscores in the rank table: the run wrote a `species_group_kmer_cov` file), with the unranked nodes below a species
    as "strain" (the run wrote a `strain_kmer_cov` file that lists them).
Expected: the run's own .summ and every .summ.<rank>_kmer_cov file, byte for byte -- including the files that exist but are
empty (the first node of a rank never gets rows) -- which also confirms the rebuilt rank table."""
import io
import json
import os
import subprocess
import tarfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "lmat_amd", "csrc", "content_summ")


def example_inputs(tmp_path):
    T = json.load(open(os.path.join(G, "example_tree.json")))
    nodes = T["nodes"]
    fastsummary = T["files"][""]
    rank = {}
    for line in fastsummary.splitlines():
        c = line.split("\t")
        rank[int(c[2])] = c[3].split(",", 1)[0].replace(" ", "_")
    parent = {n["tid"]: n["parent"] for n in nodes}
    for n in nodes:
        t = n["tid"]
        if rank.get(t, "no_rank") == "no_rank" and rank.get(parent[t]) == "species":
            rank[t] = "strain"
    children = {}
    for n in nodes:
        if n["parent"] != n["tid"]:
            children.setdefault(n["parent"], []).append(n["tid"])
    lines = ["# taxonomy tree of the example run's content_summ report", "#", str(len(nodes))]
    for n in nodes:
        ch = children.get(n["tid"], [])
        lines += [" ".join(str(x) for x in [n["tid"], len(ch)] + ch + [n["parent"]]), n["name"]]
    (tmp_path / "tax.dat").write_text("\n".join(lines))
    (tmp_path / "ranks.txt").write_text("".join(f"{n['tid']} {rank.get(n['tid'], 'no_rank')}\n" for n in nodes))
    (tmp_path / "run.fastsummary").write_text(fastsummary)
    tar = tarfile.open(os.path.join(G, "example_gene.tar.gz"))
    names = []
    for i in range(8):
        p = tmp_path / f"rl{i}.out"
        p.write_bytes(tar.extractfile(f"rl{i}.out").read())
        names.append(str(p))
    (tmp_path / "rl.flst").write_text("\n".join(names) + "\n")
    return T, rank


def test_content_summ_reproduces_the_reference_example_reports(tmp_path):
    T, rank = example_inputs(tmp_path)
    out = str(tmp_path / "run.fastsummary.summ")
    r = subprocess.run([EXE, "-c", str(tmp_path / "tax.dat"), "-l", str(tmp_path / "run.fastsummary"), "-k", "8,10,12,14,17",
                        "-f", str(tmp_path / "rl.flst"), "-r", str(tmp_path / "ranks.txt"), "-a", "plasmid,species,genus", "-o", out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert open(out).read() == T["files"][".summ"]
    want = {k: v for k, v in T["files"].items() if k.endswith("_kmer_cov")}
    got = {f[len("run.fastsummary"):]: open(os.path.join(str(tmp_path), f)).read()
           for f in os.listdir(str(tmp_path)) if f.endswith("_kmer_cov")}
    assert got == want                       # same set of files (empty ones included), same bytes
    assert len(want[".summ.species_kmer_cov"].splitlines()) == 262 and want[".summ.family_kmer_cov"] == ""
    # the stdout lines of upstream's log, up to the path names and the timer
    log = [l for l in T["files"][".summ.log"].splitlines() if not l.startswith("\t")]
    mine = r.stdout.splitlines()
    assert mine[:9] == log[:9] and mine[9].startswith("Read taxonomy tree: ") and mine[-1].startswith("query time: ")


def test_content_summ_threshold_and_rank_selection(tmp_path):
    """-v drops calls below the score; -a picks which ranks get k-mers counted (strains fold into their species, :342-351)."""
    T, rank = example_inputs(tmp_path)
    base = [EXE, "-c", str(tmp_path / "tax.dat"), "-l", str(tmp_path / "run.fastsummary"), "-f", str(tmp_path / "rl.flst"),
            "-r", str(tmp_path / "ranks.txt")]
    a = str(tmp_path / "a.summ")
    subprocess.run(base + ["-k", "10", "-a", "species", "-o", a], check=True, capture_output=True)
    b = str(tmp_path / "b.summ")
    subprocess.run(base + ["-k", "10", "-a", "species", "-v", "1.0", "-o", b], check=True, capture_output=True)
    assert open(a).read() == open(b).read() == T["files"][".summ"]           # the tree report does not depend on either
    rows = lambda fn: {l.split()[0]: l for l in open(fn) if l.startswith("taxid=")}
    ra, rb = rows(a + ".species_kmer_cov"), rows(b + ".species_kmer_cov")
    tot = lambda l: int(l.split("tot_kmer_cnt=")[1])
    assert ra.keys() == rb.keys() and "taxid=5476" in ra
    assert 0 < tot(rb["taxid=5476"]) < tot(ra["taxid=5476"])                 # fewer reads pass the threshold
    assert all("distinct_kmer_cnt=0 " in l for l in rows(a + ".genus_kmer_cov").values())  # genus not selected: nothing counted


# ---- the roll-ups of the .fastsummary (bin/run_rl.sh:251-252: tolineage.py, fsreport.py) ---------------------------------
UNCALLED_RANKS = {  # NCBI ranks of the tree's nodes that no read was called to (the .fastsummary names the rank of every called one)
    10239: "superkingdom", 35237: "no rank", 548681: "order", 860343: "no rank", 28384: "no rank", 81077: "no rank",
    2: "superkingdom", 1224: "phylum", 1236: "class", 72274: "order", 33154: "no rank", 147537: "subphylum", 4891: "class"}


def test_fastsummary_rollups_reproduce_the_reference_example(tmp_path):
    """<fastsummary>.lineage (tolineage.py: read count + ranked lineage names of every call with more than 10 reads) and
    <fastsummary>.species / .genus (fsreport.py with the gene summary gene_label wrote for the run and its threshold of 10
    reads, bin/run_gl.sh:162-164): the reference's own files for the example run, byte for byte."""
    T, rank = example_inputs(tmp_path)
    nodes = {n["tid"]: n for n in T["nodes"]}
    names_rank = {}
    for line in T["files"][""].splitlines():
        c = line.split("\t")
        names_rank[int(c[2])] = c[3].split(",", 1)[0]
    names_rank.update(UNCALLED_RANKS)
    with open(tmp_path / "names.txt", "w") as f:
        for t, n in nodes.items():
            chain = []
            x = t
            while x != 1:
                chain.append(x)
                x = nodes[x]["parent"]
            lin = "\t".join(f"{names_rank[a]},{nodes[a]['name']}" for a in reversed(chain))
            f.write(f"depth={n['depth']},taxid={t},ktaxid={t},entries=-1" + ("\t" + lin if lin else "") + "\n")
    tar = tarfile.open(os.path.join(G, "example_gene.tar.gz"))
    (tmp_path / "genesummary_tax").write_bytes(tar.extractfile("genesummary_tax").read())
    (tmp_path / "plasmids.txt").write_text("")
    exe = os.path.join(ROOT, "lmat_amd", "csrc", "fs_rollup")
    fs = str(tmp_path / "run.fastsummary")
    r = subprocess.run([exe, "-s", fs, "-u", str(tmp_path / "names.txt"), "-c", str(tmp_path / "tax.dat"), "-w", str(tmp_path / "ranks.txt"),
                        "-r", str(tmp_path / "plasmids.txt"), "-a", "plasmid,species,genus", "-g", str(tmp_path / "genesummary_tax"), "-q", "10"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert open(fs + ".lineage").read() == T["files"][".lineage"]
    assert open(fs + ".species").read() == T["files"][".species"]
    assert open(fs + ".genus").read() == T["files"][".genus"]
    assert not os.path.exists(fs + ".plasmid")          # no plasmid was called: upstream writes no file for an empty rank
    # without a gene summary: the three gene columns go
    for f in (".species", ".genus"):
        os.remove(fs + f)
    subprocess.run([exe, "-s", fs, "-c", str(tmp_path / "tax.dat"), "-w", str(tmp_path / "ranks.txt")], check=True)
    want = ["\t".join(c for i, c in enumerate(l.split("\t")) if i not in (3, 4, 5)) for l in T["files"][".species"].splitlines()]
    assert open(fs + ".species").read().splitlines() == want


def test_ordered_reports_reproduce_the_reference_example(tmp_path):
    """<fastsummary>.ordered.{plasmid,species,genus} (bin/summary.py over content_summ's report, bin/run_cs.sh:150): here made
    from OUR content_summ's files, and equal to the reference's own -- floats as Python 2 prints them, a node's k sizes in
    Python 2's dict order (8,17,10,12,14)."""
    T, rank = example_inputs(tmp_path)
    fs = str(tmp_path / "run.fastsummary")
    subprocess.run([EXE, "-c", str(tmp_path / "tax.dat"), "-l", fs, "-k", "8,10,12,14,17", "-f", str(tmp_path / "rl.flst"),
                    "-r", str(tmp_path / "ranks.txt"), "-a", "plasmid,species,genus", "-o", fs + ".summ"], check=True, capture_output=True)
    (tmp_path / "plasmids.txt").write_text("")
    exe = os.path.join(ROOT, "lmat_amd", "csrc", "fs_rollup")
    subprocess.run([exe, "-s", fs, "-c", str(tmp_path / "tax.dat"), "-w", str(tmp_path / "ranks.txt"), "-r", str(tmp_path / "plasmids.txt"),
                    "-a", "plasmid,species,genus", "-S", fs + ".summ"], check=True)
    for r in ("plasmid", "species", "genus"):
        assert open(fs + ".ordered." + r).read() == T["files"][".ordered." + r], r
