#!/usr/bin/env python3
"""End-to-end rate of the read_label-compatible CLI (reported separately from the kernel rate).

BASELINE config 1 shape scaled up in reads: ~1 M-k-mer database (tax_histo -> make_db_image), N 150 bp reads in a
FASTA file, run_rl.sh flags.  Timed by the CLI's own "Total query time" (FASTA parse + pack + H2D + kernels + D2H +
text formatting + file writes; database load excluded, as upstream's timer) and by wall clock of the process."""
import argparse, json, os, re, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2000000)
    ap.add_argument("--threads", type=int, default=8, help="-t: output shards / formatting threads")
    ap.add_argument("--keep", default=None)
    ap.add_argument("--repeat", type=int, default=1, help="the FASTA file concatenated this many times (a longer steady state without generating more reads)")
    ap.add_argument("--only", default=None, help="run only the flag set whose label contains this text")
    ap.add_argument("--db-gb", type=float, default=0.0, help="run against the synthetic table of bench.py at this size (64: the headline table) instead of the "
                    "config-1 database: built on the GPU, saved as a device image (LMATIMG2), which every CLI run then loads")
    ap.add_argument("--reps", type=int, default=1, help="runs per flag set (the JSON keeps every run and the median)")
    ap.add_argument("--fastq", action="store_true", help="also run the -p set on the same reads as a FASTQ file")
    ap.add_argument("--dir", default=None, help="where the inputs, the image and the outputs live (default: $TMPDIR; /dev/shm keeps the disk out of it)")
    a = ap.parse_args()
    from lmat_amd import synth
    d = a.keep or tempfile.mkdtemp(prefix="lmat_cli_", dir=a.dir or os.environ.get("TMPDIR", "/tmp"))
    os.makedirs(d, exist_ok=True)
    t0 = time.time()
    cache = os.path.join(d, "info.json")
    if os.path.exists(cache):  # --keep directory of an earlier run with the same --reads
        info = json.load(open(cache))
    elif a.db_gb > 0:
        # the headline table: synthetic database on the device -> device image; reads from the same generator -> FASTA
        import numpy as np
        import bench
        from lmat_amd import Engine, Params
        tax = synth.make_taxonomy(bench.BRANCHING, specials=False)
        info = synth.write_aux_files(d, tax)
        eng = Engine(0, Params.run_rl())
        eng.synth_taxonomy(bench.BRANCHING)
        tb = int(a.db_gb * (1 << 30)) // 64 * 64
        G = int(0.8 * (tb / 8) / (768 * (1.0 + 3 * (1.0 - 0.99 ** 20))))
        eng.synth_db(G, k=20, seed=2002, table_bytes=tb)
        info["n_kmers"] = eng.db_size
        info["db"] = os.path.join(d, "db.img2")
        eng.save_device_image(info["db"])
        rs = eng.synth_reads(a.reads, (150,), seed=3003)
        blob, off = rs.ascii(0, a.reads)
        rs.free()
        eng.close()
        n = a.reads
        rec = np.empty((n, 11 + 151), dtype=np.uint8)      # ">r%08d\n" (11 bytes) + 150 bases + "\n"
        rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 10] = 10; rec[:, 161] = 10
        idx = np.arange(n)
        for j in range(8):
            rec[:, 9 - j] = 48 + (idx // 10 ** j) % 10
        rec[:, 11:161] = blob[:n * 150].reshape(n, 150)
        info["fasta"] = os.path.join(d, "reads.fa")
        rec.tofile(info["fasta"])
        info["n_reads"] = n
        info = {k: v for k, v in info.items() if isinstance(v, (str, int, float))}
        json.dump(info, open(cache, "w"))
    else:
        info = synth.generate_dataset(d, (2, 2, 2, 2, 3, 3), 13400, a.reads, L=150)
        info = {k: v for k, v in info.items() if isinstance(v, (str, int, float))}
        info["n_reads"] = a.reads
        json.dump(info, open(cache, "w"))
    gen_s = time.time() - t0
    if a.repeat > 1:
        big = os.path.join(d, f"reads_x{a.repeat}.fa")
        if not os.path.exists(big):
            with open(big, "wb") as o:
                blob = open(info["fasta"], "rb").read()
                for _ in range(a.repeat):
                    o.write(blob)
        info["fasta"] = big
        info["n_reads"] = info["n_reads"] * a.repeat
    csrc = os.path.join(ROOT, "lmat_amd", "csrc")
    if a.db_gb > 0:
        img = info["db"]
    else:
        img = os.path.join(d, "db.img")
        subprocess.run([os.path.join(csrc, "make_db_image"), "-i", info["db"], "-o", img, "-k", "20", "-f", info["idmap"]],
                       check=True, stdout=subprocess.DEVNULL)
    fq = None
    if a.fastq:  # the same reads as 4-line FASTQ records
        fq = os.path.join(d, "reads.fq")
        if not os.path.exists(fq):
            with open(info["fasta"], "rb") as f, open(fq, "wb") as o:
                qual = None
                while True:
                    h = f.readline()
                    if not h:
                        break
                    sq = f.readline()
                    if qual is None or len(qual) != len(sq):
                        qual = b"I" * (len(sq) - 1) + b"\n"
                    o.write(b"@" + h[1:] + sq + b"+\n" + qual)
    try:
        host_cpus = int(open("/sys/fs/cgroup/cpu.max").read().split()[0]) / int(open("/sys/fs/cgroup/cpu.max").read().split()[1])
    except Exception:
        host_cpus = os.cpu_count()
    out = {"reads": info["n_reads"], "db_kmers": info["n_kmers"], "db_gib": a.db_gb or None, "generate_s": round(gen_s, 1), "host_cpus": host_cpus,
           "directory": d, "runs": [], "median": {}}
    def throttled():
        try:
            return dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        except OSError:
            return {}
    sets = [("-p, reads echoed", ["-p"], info["fasta"]), ("-p -a (calls + candidates, no echo)", ["-p", "-a"], info["fasta"]), ("-a (calls only)", ["-a"], info["fasta"])]
    if fq:
        sets.append(("-p -q, FASTQ input, reads echoed", ["-p", "-q"], fq))
    for label, extra, query in [x for x in sets for _ in range(max(a.reps, 1))]:
        if a.only and a.only not in label:
            continue
        th0 = throttled()
        cmd = [os.path.join(csrc, "read_label"), "-f", info["idmap"], "-u", info["names"], "-w", info["rank"], "-x", "0", "-j", "30",
               "-l", "0", "-b", "1.0", "-e", info["depth"], "-t", str(a.threads), "-i", query, "-d", img,
               "-c", info["tree"], "-o", os.path.join(d, "out")] + extra
        t0 = time.time()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, LMAT_CLI_TIMING="1"))
        wall = time.time() - t0
        if p.returncode != 0:
            print(p.stdout[-2000:], file=sys.stderr)
            raise SystemExit(p.returncode)
        m = re.search(r"Total query time: ([0-9.eE+-]+) sec", p.stdout)
        q = float(m.group(1)) if m else None
        size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.startswith("out") and f.endswith(".out"))
        tm = re.search(r"\[read_label\] (stage.*)", p.stdout)
        tl = re.search(r"\[read_label\] (timeline.*)", p.stdout)
        th1 = throttled()
        out["runs"].append({"cfs_throttled_periods": int(th1.get("nr_throttled", 0)) - int(th0.get("nr_throttled", 0)),
                            "cpu_s": (int(th1.get("usage_usec", 0)) - int(th0.get("usage_usec", 0))) / 1e6, "flags": label, "split": tm.group(1) if tm else None, "timeline": tl.group(1) if tl else None, "query_s": q, "wall_s": round(wall, 2), "reads_per_s_query": round(info["n_reads"] / q) if q else None,
                            "reads_per_s_wall": round(info["n_reads"] / wall), "out_bytes": size})
    import statistics
    for label in dict.fromkeys(r["flags"] for r in out["runs"]):
        qs = [r["reads_per_s_query"] for r in out["runs"] if r["flags"] == label and r["reads_per_s_query"]]
        if qs:
            out["median"][label] = {"reads_per_s_query": statistics.median(qs), "runs": len(qs), "min": min(qs), "max": max(qs),
                                    "reads_per_s_per_host_cpu": statistics.median(qs) / host_cpus if host_cpus else None}
    for f in os.listdir(d):   # the outputs of the last run: gigabytes
        if f.startswith("out") and (f.endswith(".out") or "summary" in f or f.endswith("sum")):
            os.unlink(os.path.join(d, f))
    if not a.keep:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
