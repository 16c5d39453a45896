#!/usr/bin/env python3
"""Builds tests/golden/run_rl_argv.json: the argv that the reference's own driver script, bin/run_rl.sh, hands to
`read_label` for a set of command lines (bin/run_rl.sh:88-243: option spellings, FASTQ auto-detection :162-177, the
pruning branch :209-211, the three null-model branches :215-228).

How: run_rl.sh is run here, in the build container, with a RECORDING STUB named read_label first on PATH (it writes its
arguments, one per line, and touches the .fastsummary the script waits for; tolineage.py / fsreport.py are stubs too).
The script is executed from a temporary copy in which the prefix "/usr/bin/time -v " of the read_label line is dropped,
because this image has no /usr/bin/time; nothing else is changed and nothing of the script is kept.
The fixture is DATA: option list in, argv out, with the temporary paths replaced by $LMAT_DIR / $QUERY / $DB / $ODIR."""
import json
import os
import subprocess
import sys
import tempfile

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/bin/run_rl.sh"
here = os.path.dirname(os.path.abspath(__file__))
cases = [
    {"name": "fasta_no_nullmodel", "fastq": False, "opts": ["--nullm=no", "--threads=4"]},
    {"name": "fastq_autodetect", "fastq": True, "opts": ["--nullm=no", "--threads=2"]},
    {"name": "default_null_models", "fastq": False, "opts": ["--threads=3"]},
    {"name": "explicit_null_models", "fastq": False, "opts": ["--nullm=$LMAT_DIR/my_null_lst.txt", "--threads=1"]},
    {"name": "pruning", "fastq": False, "opts": ["--nullm=no", "--prune_thresh=5", "--threads=2"]},
    {"name": "all_knobs", "fastq": False, "opts": ["--nullm=no", "--threads=8", "--sdiff=0.5", "--hbias=1.5", "--min_score=0.25",
                                                     "--min_read_kmer=40", "--verbose", "--overwrite"]},
]
out = []
with tempfile.TemporaryDirectory() as td:
    lmat_dir, bindir, odir = os.path.join(td, "lmatdir"), os.path.join(td, "bin"), os.path.join(td, "out")
    for d in (lmat_dir, bindir, odir):
        os.makedirs(d)
    script = os.path.join(td, "run_rl.sh")
    text = open(ref).read()
    assert text.count("/usr/bin/time -v $rprog") == 1
    open(script, "w").write(text.replace("/usr/bin/time -v $rprog", "$rprog"))
    rec = os.path.join(td, "argv.txt")
    stub = os.path.join(bindir, "read_label")
    open(stub, "w").write('#!/bin/bash\nprintf "%s\\n" "$@" > ' + rec + '\nwhile [ $# -gt 0 ]; do case "$1" in -o) o="$2";; -x) x="$2";; -j) j="$2";; esac; shift; done\n: > "$o.$x.$j.fastsummary"\n')
    for name in ("tolineage.py", "fsreport.py"):
        open(os.path.join(bindir, name), "w").write("#!/bin/bash\nexit 0\n")
    for f in os.listdir(bindir):
        os.chmod(os.path.join(bindir, f), 0o755)
    db = os.path.join(td, "some.db")
    open(db, "w").write("x")
    open(os.path.join(lmat_dir, "my_null_lst.txt"), "w").write("")
    for c in cases:
        q = os.path.join(td, "reads.fq" if c["fastq"] else "reads.fna")
        open(q, "w").write("@r1\nACGT\n+\nIIII\n" if c["fastq"] else ">r1\nACGT\n")
        if os.path.exists(rec):
            os.remove(rec)
        for f in os.listdir(odir):  # the script skips a query whose .fastsummary exists
            os.remove(os.path.join(odir, f))
        opts = [o.replace("$LMAT_DIR", lmat_dir) for o in c["opts"]]
        env = dict(os.environ, LMAT_DIR=lmat_dir, PATH=bindir + ":" + os.environ["PATH"])
        subprocess.run(["bash", script, "--db_file=" + db, "--query_file=" + q, "--odir=" + odir] + opts, env=env,
                       stdout=subprocess.DEVNULL if os.environ.get("QUIET", "1") == "1" else None, stderr=subprocess.STDOUT)
        argv = [l.rstrip("\n") for l in open(rec)]
        sub = lambda s: s.replace(lmat_dir, "$LMAT_DIR").replace(odir, "$ODIR").replace(q, "$QUERY").replace(db, "$DB").replace(
            os.path.basename(q), "$QUERYNAME").replace(os.path.basename(db), "$DBNAME")
        out.append({"name": c["name"], "fastq": c["fastq"], "run_rl_options": ["--db_file=$DB", "--query_file=$QUERY", "--odir=$ODIR"] + c["opts"],
                    "read_label_argv": [sub(a) for a in argv]})
json.dump(out, open(os.path.join(here, "run_rl_argv.json"), "w"), indent=1)
for c in out:
    print(c["name"], " ".join(c["read_label_argv"]))
