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
    a = ap.parse_args()
    from lmat_amd import synth
    d = a.keep or tempfile.mkdtemp(prefix="lmat_cli_", dir=os.environ.get("TMPDIR", "/tmp"))
    os.makedirs(d, exist_ok=True)
    t0 = time.time()
    cache = os.path.join(d, "info.json")
    if os.path.exists(cache):  # --keep directory of an earlier run with the same --reads
        info = json.load(open(cache))
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
    img = os.path.join(d, "db.img")
    subprocess.run([os.path.join(csrc, "make_db_image"), "-i", info["db"], "-o", img, "-k", "20", "-f", info["idmap"]],
                   check=True, stdout=subprocess.DEVNULL)
    out = {"reads": info["n_reads"], "db_kmers": info["n_kmers"], "generate_s": round(gen_s, 1), "runs": []}
    def throttled():
        try:
            return dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        except OSError:
            return {}
    for label, extra in (("-p, reads echoed", ["-p"]), ("-p -a (calls + candidates, no echo)", ["-p", "-a"]), ("-a (calls only)", ["-a"])):
        if a.only and a.only not in label:
            continue
        th0 = throttled()
        cmd = [os.path.join(csrc, "read_label"), "-f", info["idmap"], "-u", info["names"], "-w", info["rank"], "-x", "0", "-j", "30",
               "-l", "0", "-b", "1.0", "-e", info["depth"], "-t", str(a.threads), "-i", info["fasta"], "-d", img,
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
    print(json.dumps(out))


if __name__ == "__main__":
    main()
