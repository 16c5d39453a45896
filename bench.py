#!/usr/bin/env python3
"""bench.py -- reads/s of the read-labeling hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the classify path (k-mer extraction -> hash probe -> taxid
accumulation -> LCA/score call) over one batch of synthetic 150 bp reads that are
already packed in HBM, against a synthetic k-mer DB of --db-gb GiB generated on the
device.  One process per GPU; the DB is replicated, reads are
sharded (weak scaling: per-GPU batch fixed); the only collective is the final
all-reduce of the per-taxid tallies (RCCL).

Prints ONE JSON line on rank 0 with `roofline` (HBM, algorithmic bytes per launch /
HIP-event kernel time) and, at N=1, `cpu_baseline` (the CPU oracle on a bounded
sample of the same reads, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BRANCHING = (3, 4, 4, 4, 4, 3)  # 768 species x 3 strains, 3328 nodes


def sample_algorithmic_bytes(eng, reads, n_sample, k):
    """Mean algorithmic bytes per read, SURVEY.md 8(d):
    ceil(L/4)+ceil(L/8) + 64*D + sum over probed k-mers with a non-inline list of 64*ceil(2*len/64) + 16."""
    from lmat_amd import synth
    blob, off = reads.ascii(0, n_sample)
    lut = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    total = 0.0
    bucket_only = 0.0
    all_k, per_read = [], []
    for i in range(n_sample):
        seq = blob[int(off[i]):int(off[i + 1])]
        L = seq.size
        b = np.ceil(L / 4) + np.ceil(L / 8) + 16
        codes = lut[seq]
        if L >= k:
            bad = codes == 255
            c2 = np.where(bad, 0, codes)
            km = synth.kmers_of(c2, k)
            win_bad = np.convolve(bad.astype(np.int32), np.ones(k, dtype=np.int32), mode="valid") > 0
            km = np.unique(km[~win_bad])
            if km.size >= eng.params.min_kmer or True:
                all_k.append(km)
                per_read.append((i, km.size))
                b += 64.0 * km.size
        total += b
        bucket_only += b
    probes = None
    if all_k:
        cat = np.concatenate(all_k)
        counts, _ = eng.lookup(cat, stride=1)
        multi = counts[counts > 1].astype(np.float64)
        total += float(np.sum(64.0 * np.ceil(2.0 * multi / 64.0)))
        probes = eng.probe_stats(cat)   # where these lookups end: the share that needs the overflow table is a second dependent request
        n_all = max(sum(probes[x] for x in ("home_hit", "absent_one_request", "overflow_hit", "overflow_miss")), 1)
        probes["lookups"] = n_all
        probes["overflow_probe_share"] = (probes["overflow_hit"] + probes["overflow_miss"]) / n_all
        probes["hits_found_in_overflow_share"] = probes["overflow_hit"] / max(probes["overflow_hit"] + probes["home_hit"], 1)
    return total / n_sample, bucket_only / n_sample, probes


def cpu_info():
    """CPU model, logical CPUs, SMT state and the CPUs this process may use (cgroup quota) of the host."""
    model, smt = "unknown", "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        smt = "on" if open("/sys/devices/system/cpu/smt/active").read().strip() == "1" else "off"
    except OSError:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        pass
    return {"model": model, "logical_cpus": os.cpu_count(), "smt": smt, "cgroup_cpu_quota": quota}


def cpu_baseline(eng, reads, n_sample, k, tmpdir):
    """CPU oracle (oracle/, a port of read_label's proc_line) on the first n_sample reads of this rank's
    workload, at T=1 and at T=the CPUs this process may use.  Its k-mer table holds exactly the taxid lists the GPU
    table returns for those reads' k-mers."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    from lmat_amd import synth
    info = cpu_info()
    threads = int(info["cgroup_cpu_quota"] or info["logical_cpus"] or 1)
    tax = synth.make_taxonomy(BRANCHING, specials=False)
    p = synth.write_aux_files(tmpdir, tax)
    orc = oracle_py.Oracle(p["tree"], p["depth"], p["rank"], p["idmap"])
    orc.set_k(k)
    orc.set_options(sdiff=eng.params.sdiff, hbias=eng.params.hbias, prn_all=eng.params.prn_all, min_kmer=eng.params.min_kmer)
    blob, off = reads.ascii(0, n_sample)
    kms = []
    for i in range(n_sample):
        km, _, _, _ = orc.extract(bytes(blob[int(off[i]):int(off[i + 1])]), k)
        kms.append(km)
    kms = np.unique(np.concatenate(kms))
    counts, tids = eng.lookup(kms, stride=32)
    orc.add_lists32(kms, counts, tids)
    b0 = np.append(blob, np.uint8(0))

    def rate(t, n, budget_s):
        o = off[:n + 1]
        t0 = time.perf_counter()
        orc.classify_mt(b0, o, k, t)
        d1 = time.perf_counter() - t0
        reps = int(min(max(budget_s / max(d1, 1e-3), 1), 400))
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.classify_mt(b0, o, k, t)
        dt = time.perf_counter() - t0
        return n * reps / dt, dt, reps

    n1 = max(n_sample // 24, 1000)
    r1, s1, _ = rate(1, n1, 5.0)                # T = 1 on a slice of the sample (~5 s)
    rn, sn, reps = rate(threads, n_sample, 10.0)  # T = N (~10 s)
    orc.close()
    # the host table: one unordered_map node per k-mer (key + vector header + link + allocator overhead, ~64 B) plus the heap
    # block of its list (>= 32 B): what a pass over the sample walks, against the caches of the host
    n_tab = int(kms.size)
    table_bytes = n_tab * 96.0
    ratio, ratio_from = None, None   # oracle / reference on the reference's own retrieve_kmer_labels, measured in the build container
    return {"value": rn, "unit": "reads/s", "cores": threads, "kind": "port", "t1_reads_per_s": r1, "tN_reads_per_s": rn,
            "port_over_reference": ratio, "reference_equivalent_reads_per_s": rn / ratio if ratio else None, "port_over_reference_from": ratio_from,
            "cpu": info,
            "table": f"host-resident std::unordered_map of {n_tab} k-mers with their taxid lists, ~{table_bytes / 2**30:.2f} GiB of nodes and list blocks "
                     "(every pass walks all of it: larger than the L2/L3 a thread sees, so lookups miss to DRAM; the 64 GiB table itself is not copied to the host)",
            "sample": f"first {n_sample} reads of the same synthetic workload x {reps} passes at T={threads} ({sn:.1f} s), first {n1} reads at "
                      f"T=1 ({s1:.1f} s); CPU oracle (oracle/lmat_oracle.hpp, a restatement of read_label's proc_line -- the reference's "
                      "read_label.cpp does not build in this image; port_over_reference is for the part that does); k-mer table = host hash "
                      "map holding the GPU table's lists for those reads' k-mers"}


def e2e_stream(eng, reads, batch, steps, warmup, log):
    """The boundary-inclusive rate (first batch H2D start -> last result D2H complete): ASCII reads in pinned
    host memory -> H2D -> pack -> classify -> D2H of the 40-byte results, through the ring of pinned slots of
    lmat_stream_* (3 slots, copy in / kernels / copy out of consecutive batches overlapped on three HIP streams)."""
    import ctypes as C
    from lmat_amd import Stream
    from lmat_amd.capi import host_alloc, host_free
    nbuf = 4
    bufs, offs = [], []
    for j in range(nbuf):
        blob, off = reads.ascii(j * batch, batch)
        addr = host_alloc(blob.size + 16)
        C.memmove(addr, blob.ctypes.data, blob.size)
        bufs.append(addr)
        offs.append(np.ascontiguousarray(off, dtype=np.uint64))
    max_bases = max(int(o[-1]) for o in offs) + 16
    st = Stream(eng, batch, max_bases, cands_per_read=0, n_slots=3)
    import torch

    def run(n):
        for i in range(n):
            if st.in_flight == st.n_slots:
                st.next_nocopy()
            st.submit_pinned(bufs[i % nbuf], offs[i % nbuf], batch, tag=i)
        while st.in_flight:
            st.next_nocopy()

    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st.close()
    for a in bufs:
        host_free(a)
    bytes_in = sum(int(offs[i % nbuf][-1]) + 16 * (batch + 1) for i in range(steps))  # bases + byte offsets + record offsets
    bytes_out = steps * batch * 40
    log(f"boundary-inclusive stream: {steps} x {batch} reads in {dt:.3f}s")
    return {"value": steps * batch / dt, "unit": "reads/s", "ms_per_step": dt / steps * 1e3,
            "h2d_GBs": bytes_in / dt / 1e9, "d2h_GBs": bytes_out / dt / 1e9,
            "what": "ASCII reads in pinned host memory -> H2D -> pack -> classify -> D2H of the 40-B results per step, 3 pinned slots, "
                    "copy in / kernels / copy out overlapped (lmat_stream_*); calls only"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes, one per GPU, started by THIS process, which
    never imports torch or touches HIP itself (a parent that had could not safely start GPU children).  Each child gets the
    environment a launcher would set; rank 0's stdout (the JSON line) is this process's stdout.  The children are watched
    together: the first one that exits non-zero (out of memory, a HIP error) takes the others -- which would wait for it in a
    collective for ever -- down with it; LMAT_BENCH_TIMEOUT_S bounds the whole run."""
    import socket
    import subprocess
    with socket.socket() as so:  # (a free port at this moment; rank 0 binds it again a few seconds later)
        so.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + float(os.environ.get("LMAT_BENCH_TIMEOUT_S", "3000"))
    rc, live = 0, list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = max(rc, abs(code))
        if (rc or time.monotonic() > deadline) and live:
            if not rc:
                rc = 124
                print(f"bench.py: ranks still running after LMAT_BENCH_TIMEOUT_S: stopping them", file=sys.stderr)
            for p in live:   # exactly the processes started above
                p.terminate()
            t_end = time.monotonic() + 10
            for p in live:
                try:
                    p.wait(timeout=max(0.1, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            live = []
        if live:
            time.sleep(0.05)
    return rc


def plan_steps(args, world):
    """Reads per step and per launch of one rank.  Weak scaling (default): --batch reads per rank and step whatever the
    number of ranks.  --total-reads N (strong scaling, BASELINE config 4 "50M reads sharded across 8"): the job's N reads are
    split evenly over the ranks -- contiguous shares, as read_label splits its input -- and a rank's share over --steps steps.
    A step is a whole number of launches of --launch-reads reads (args.batch is rounded up to that).  -> strong?"""
    strong = args.total_reads > 0
    if strong:
        per_rank = (args.total_reads + world - 1) // world
        args.batch = max(1, (per_rank + args.steps - 1) // args.steps)
    args.launch_reads = max(1, min(args.launch_reads, args.batch))
    lps = (args.batch + args.launch_reads - 1) // args.launch_reads
    if strong:  # even launches: the step's reads split over its launches instead of a last short one
        args.launch_reads = (args.batch + lps - 1) // lps
    args.batch = lps * args.launch_reads
    return strong


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--db-gb", type=float, default=64.0, help="hash table size in GiB (BASELINE metric: 64)")
    ap.add_argument("--batch", type=int, default=8_000_000, help="reads per step per GPU (rounded up to whole launches)")
    ap.add_argument("--launch-reads", type=int, default=8_000_000, help="reads per classify launch: a step is batch / launch-reads launches (default: one launch per step; rounds 1-3 quoted launches of 2 M: --launch-reads 2000000)")
    ap.add_argument("--read-len", default="150", help="read length, or a comma list for a mixed-length batch (not the headline workload)")
    ap.add_argument("--cpu-sample", type=int, default=120000, help="reads of the CPU baseline's sample (120 k reads = 15.7 M distinct k-mers = ~1.4 GiB of host table: DRAM-missing)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--list-replicas", type=int, default=1, help="copies of every distinct taxid list in the arena, one per 512-base stretch of a genome (5: 4 M lists, 250 MB -- the most 24-bit payloads address)")
    ap.add_argument("--genus-permille", type=int, default=100, help="share of every genome that is a block shared within its genus (SURVEY 8d: 100)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the boundary-inclusive (pinned host -> H2D -> classify -> D2H) leg")
    ap.add_argument("--windows", type=int, default=5, help="the timed window of --steps steps is `value`; this many further windows of the same steps give value_median_of_windows (0: none)")
    ap.add_argument("--as-ranks", type=int, default=1, help="tests: ONE rank classifies the read sets of this many ranks one after the other (what an N-rank run must add up to)")
    ap.add_argument("--tally-out", default=None, help="tests: rank 0 writes the merged per-taxid tallies to this JSON file")
    ap.add_argument("--genome-len", type=int, default=0, help="bases per strain genome (default: sized so that the table holds ~6.4 k-mers per bucket)")
    ap.add_argument("--list-tail", default="0,0,0", help="heavy tail of taxid lists: thousandths of every genome that are blocks conserved across a whole family, phylum, superkingdom (lists of 69 / 277 / 1109 taxids): e.g. 20,10,5")
    ap.add_argument("--prune", type=int, default=0, help="run-time pruning -g N with a numeric rank map (-m): lists longer than N lose their lowest ranks first (TaxNodeStat.hpp:76-203)")
    ap.add_argument("--total-reads", type=int, default=0, help="STRONG scaling (BASELINE config 4: `50M reads sharded across 8`): this many reads in total, split evenly over the ranks and over --steps steps (overrides --batch; `scaling` becomes \"strong\")")
    ap.add_argument("--no-cands", action="store_true", help="skip the value_with_candidates leg (the same window with -p: candidate pairs written, as bin/run_rl.sh runs it)")
    ap.add_argument("--cands-per-read", type=int, default=32, help="capacity of the candidate buffer of the -p leg, pairs per read")
    ap.add_argument("--spawn-check", action="store_true", help="tests: every rank prints the environment it was started with and exits before touching a GPU")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))

    T0 = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.spawn_check:
        if os.environ.get("LMAT_SPAWN_CHECK_FAIL") == str(rank):   # tests: this rank dies, the others would wait for ever
            sys.exit(3)
        if os.environ.get("LMAT_SPAWN_CHECK_FAIL") is not None:
            time.sleep(120)
        print(json.dumps({"rank": rank, "world": world, "local_rank": local_rank, "master": os.environ.get("MASTER_ADDR"),
                          "torch_loaded_in_parent": "torch" in sys.modules}), file=sys.stderr if rank else sys.stdout)
        return
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    import torch
    # LMAT_BENCH_REHEARSE=1: N ranks on ONE GPU over gloo, to rehearse the multi-rank control flow on a 1-GPU box
    rehearse = os.environ.get("LMAT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from lmat_amd import Engine, Params
    k = 20

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - T0:.1f}s] {msg}", file=sys.stderr, flush=True)

    eng = Engine(local_rank, Params.run_rl(prn_all=0))  # run_rl.sh flags, calls-only output
    eng.synth_taxonomy(BRANCHING)
    tail = tuple(int(x) for x in args.list_tail.split(","))
    if args.prune > 0:  # -g N -m <numeric ranks>: the rank value of a synthetic taxid is its level (strains 6: pruned first)
        import tempfile as _tf
        with _tf.NamedTemporaryFile("w", suffix=".ranks", delete=False) as f:
            f.write("1 0\n")
            dense, lv_n = 0, 1
            for lv, b in enumerate(BRANCHING):
                lv_n *= b
                for _ in range(lv_n):
                    dense += 1
                    f.write(f"{1000 + 7 * dense} {lv + 1}\n")
            rank_fn = f.name
        eng.set_label_modes(False, args.prune, rank_fn)
    table_bytes = int(args.db_gb * (1 << 30)) // 64 * 64
    n_species, S = 768, 3
    pm = 1.0 - 0.99 ** k
    G = args.genome_len or int(0.8 * (table_bytes / 8) / (n_species * (1.0 + S * pm)))
    t0 = time.perf_counter()
    eng.synth_db(G, k=k, seed=2002, table_bytes=table_bytes, genus_block_permille=args.genus_permille, list_replicas=args.list_replicas, conserved_permille=tail)
    t_build = time.perf_counter() - t0
    log(f"db built: {eng.db_size} k-mers in {t_build:.2f}s, table {table_bytes / 2**30:.1f} GiB, G={G}")
    strong = plan_steps(args, world)
    lps = args.batch // args.launch_reads  # launches per step
    n_steps_total = args.steps + args.warmup
    n_reads = args.batch * n_steps_total
    read_lens = tuple(int(x) for x in str(args.read_len).split(","))
    args.read_len = read_lens[0] if len(read_lens) == 1 else "/".join(str(x) for x in read_lens)
    # rank r's reads come from seed 3003 + 7919 r; --as-ranks R (tests): this one rank takes the sets of ranks 0..R-1 in turn
    read_sets = [eng.synth_reads(n_reads, read_lens, seed=3003 + 7919 * (rank + j)) for j in range(args.as_ranks)]
    reads = read_sets[0]
    log(f"{n_reads} reads generated ({reads.device_bytes / 2**20:.0f} MiB packed)")
    # the tally merge across ranks is the engine's own RCCL all-reduce (lmat_comm_*, collective.cpp): rank 0 makes the
    # communicator id, the launcher's channel (here: torch.distributed's broadcast) hands it round.  The rehearsal mode
    # (N ranks on ONE GPU) cannot form an RCCL communicator -- two ranks on one device -- and sums over gloo instead.
    merge_path = "none"
    if dist is not None and not rehearse:
        # Should the engine's communicator not come up on some rank (a library that does not load, an IPC setting), every rank
        # falls back to the same all-reduce through torch.distributed (backend "nccl" = RCCL as well) on the same buffers in
        # HBM: the run still measures the path; the line says which merge it used.
        # Pre-flight first, so that no rank can enter ncclCommInitRank while another never will: every rank checks that
        # librccl and its entry points load (rank 0 also makes the id), all ranks agree on that (MIN), and only then is the id
        # broadcast -- by ALL ranks, whatever happened -- and the communicator made.
        uid = torch.zeros(128, dtype=torch.uint8, device=f"cuda:{local_rank}")
        ok = 1
        try:
            if not Engine.comm_available():
                raise RuntimeError("librccl or one of its entry points did not load")
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(Engine.comm_unique_id()), dtype=torch.uint8))
        except Exception as e:  # noqa: BLE001
            ok = 0
            print(f"[bench] rank {rank}: engine communicator unavailable: {e}", file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{local_rank}")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            dist.broadcast(uid, src=0)
            try:
                eng.comm_init(bytes(uid.cpu().numpy().tobytes()), world, rank)
            except Exception as e:  # noqa: BLE001
                ok = 0
                print(f"[bench] rank {rank}: lmat_comm_init failed: {e}", file=sys.stderr, flush=True)
            flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{local_rank}")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        merge_path = "engine_rccl" if int(flag.item()) == 1 else "torch_rccl"
        log(f"tally merge across {world} ranks: {merge_path}")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # tallies exposed to torch for the RCCL all-reduce (no copy)
    n_ids, nbytes = eng.counts_layout()

    class _Buf:
        def __init__(self, ptr, n, typestr):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}

    base = eng.counts_device_ptr()
    t_cnt = torch.as_tensor(_Buf(base, n_ids, "<i8"), device=f"cuda:{local_rank}")
    t_sc = torch.as_tensor(_Buf(base + 8 * n_ids, n_ids, "<f8"), device=f"cuda:{local_rank}")
    t_nm = torch.as_tensor(_Buf(base + 16 * n_ids, 3, "<i8"), device=f"cuda:{local_rank}")

    def run_step(rs, step):
        for l_ in range(lps):
            eng.classify_async(rs, (step * lps + l_) * args.launch_reads, args.launch_reads)

    def reduce_max(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    for w in range(args.warmup):
        run_step(reads, w)
    wms, wl = eng.sync()
    log(f"warmup done: {wl} launches, {wms:.1f} ms kernel time")
    eng.counts_reset()
    barrier()
    t0 = time.perf_counter()
    for rs in read_sets:
        for s in range(args.steps):
            run_step(rs, args.warmup + s)
    kernel_ms, launches = eng.sync()
    classify_ms, decide_ms, _ = eng.last_timing()
    class_flow = eng.last_counters()   # reads the last launch passed from capacity class to capacity class
    from lmat_amd.shard import allreduce_tallies
    if rehearse and dist is not None:  # gloo: stage through host copies
        host = [x.cpu() for x in (t_cnt, t_sc, t_nm)]
        allreduce_tallies(host[0], host[1], host[2], dist)
        for d_, h_ in zip((t_cnt, t_sc, t_nm), host):
            d_.copy_(h_)
    elif dist is not None and merge_path == "engine_rccl":
        eng.comm_allreduce_counts()  # merge step of read_label.cpp:1760-1800: ncclAllReduce on the tally buffers in HBM
    elif dist is not None:
        allreduce_tallies(t_cnt, t_sc, t_nm, dist)  # the same sums over torch.distributed (RCCL), in place in HBM
    barrier()
    dt = reduce_max(time.perf_counter() - t0)

    log(f"timed region {dt:.3f}s, kernels {kernel_ms:.1f} ms over {launches} launches (classify {classify_ms:.1f} + decide {decide_ms:.1f})")
    total_reads = args.batch * args.steps * world * args.as_ranks
    value = total_reads / dt
    merged = eng.counts() if rank == 0 else None   # the job's tallies, before the extra windows add to them
    # further windows of the same K steps (same barriers, max over ranks): their median says how far one slow launch moved `value`
    win = []
    for _ in range(max(args.windows, 0)):
        barrier()
        t0 = time.perf_counter()
        for s in range(args.steps):
            run_step(reads, args.warmup + s)
        eng.sync()
        barrier()
        win.append(args.batch * args.steps * world / reduce_max(time.perf_counter() - t0))
    # The same window as `value`, as bin/run_rl.sh:243 runs read_label: with -p, every read's candidate (taxid, score) pairs
    # written next to its record (8 bytes per pair on top of the algorithmic bytes).  Same steps, same barriers.
    with_cands = None
    if not args.no_cands:
        eng.set_params(Params.run_rl(prn_all=1))
        cap = args.launch_reads * max(args.cands_per_read, 1)

        def run_step_c(step):
            for l_ in range(lps):
                eng.classify_async_cands(reads, (step * lps + l_) * args.launch_reads, args.launch_reads, cap)

        run_step_c(0)
        eng.sync()
        barrier()
        t0 = time.perf_counter()
        for s in range(args.steps):
            run_step_c(args.warmup + s)
        c_kernel_ms, c_launches = eng.sync()
        c_classify_ms, _, _ = eng.last_timing()
        barrier()
        c_dt = reduce_max(time.perf_counter() - t0)
        last_recs = eng.fetch_results(0, min(args.launch_reads, 1_000_000))   # records of the last launch: pairs written per read
        with_cands = {"dt": c_dt, "classify_ms": c_classify_ms, "launches": c_launches, "pairs_per_read": float(last_recs["n_cand"].mean())}
        eng.set_params(Params.run_rl(prn_all=0))
    if rank == 0:
        counts, nomatch = merged
        called = sum(c for c, _ in counts.values())
        if args.tally_out:
            with open(args.tally_out, "w") as f:
                json.dump({"n_gpus": world, "counts": {str(t): c for t, (c, _) in sorted(counts.items())},
                           "scores": {str(t): sc for t, (_, sc) in sorted(counts.items())}, "nomatch": nomatch}, f)
        mean_b, mean_bucket, probes = sample_algorithmic_bytes(eng, reads, 2000, k)
        log(f"algorithmic bytes/read = {mean_b:.0f} (reads + one 64-B bucket per distinct k-mer + result: {mean_bucket:.0f})")
        avg_ms = classify_ms / max(launches, 1)  # dominant kernel only: classify_kernel (HBM-bound probe inside)
        achieved = mean_b * args.launch_reads / (avg_ms * 1e-3) / 1e9
        # every kernel of a launch: the classify kernel and what follows it (re-run classes, general decision path).  Those run on
        # side streams beside the NEXT launch's classify kernel, so their event times overlap it and do not add up: the wall time of
        # the timed region per launch is the honest figure
        step_ms = dt * 1e3 / max(args.steps * args.as_ranks * lps, 1)
        # HBM bytes per launch come from PMC counters, which only a rocprofv3 run of this command can collect
        # (scripts/profile_gpu.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes, condensed into profiles/): the number
        # is carried over from that committed file and labelled as such, never presented as measured in this run
        traffic, traffic_from, frac_measured = None, None, None
        tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                same = tj.get("reads_per_launch") == args.launch_reads and tj.get("db_gib") == args.db_gb and tj.get("read_len") == args.read_len
                if same:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_from = f"{tj.get('from')}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on another run (kernel {tj.get('kernel_avg_ms'):.2f} ms there)"
                    frac_measured = traffic / (tj.get("kernel_avg_ms") * 1e-3) / 1e9 / HBM_PEAK_GBS
            except Exception:
                traffic = None
        g_ms, g_bytes = eng.gather_bench(1 << 28)
        gather_gbs = g_bytes / (g_ms * 1e-3) / 1e9
        log(f"random 64-B gather ceiling on this table: {gather_gbs:.0f} GB/s")
        out = {
            "metric": "reads/s (150 bp) vs 64 GB k-mer DB", "value": value, "unit": "reads/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / (args.steps * args.as_ranks) * 1e3,
            "value_median_of_windows": float(np.median(win)) if win else None, "windows": [round(x) for x in win],
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "tally_merge": ("gloo_rehearsal" if rehearse and dist is not None else merge_path), "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.batch * args.steps} x {args.read_len} bp reads/GPU vs {args.db_gb:g} GiB "
                                   f"k-mer hash ({eng.db_size} 20-mers, replicated per GPU), run_rl.sh flags -x 0 -j 30 -l 0 -b 1, calls-only",
                       "total_reads": args.batch * args.steps * world * args.as_ranks, "reads_per_step_per_gpu": args.batch, "launches_per_step": lps, "reads_per_launch": args.launch_reads, "read_len": args.read_len, "db_gib": args.db_gb,
                       "db_kmers": eng.db_size, "k": k, "parallelism": f"reads sharded x{world}, DB replicated",
                       "db_build_s": round(t_build, 2), "genus_block_permille": args.genus_permille, "list_tail_permille": list(tail), "prune_g": args.prune,
                       "reads_past_each_class_last_launch": dict(class_flow, reads_per_launch=args.launch_reads), "list_replicas": args.list_replicas, "distinct_lists": eng.n_lists, "list_arena_mib": round(eng.arena_bytes / 2**20, 1), "reads_called": called, "nomatch": nomatch},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_from": traffic_from,
                         "frac_measured": frac_measured,
                         "frac_step": mean_b * args.launch_reads / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_bucket_only": mean_bucket * args.launch_reads / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_read_bucket_only": mean_bucket,
                         "kernel": "classify_kernel<160,64,256,false,false,true,false>",
                         "kernel_avg_ms": avg_ms, "kernel_ms_per_2M_reads": avg_ms * 2e6 / args.launch_reads, "step_ms_per_launch": step_ms, "step_ms_per_2M_reads": step_ms * 2e6 / args.launch_reads, "tail_kernels_event_ms": decide_ms / max(launches, 1), "algorithmic_bytes_per_read": mean_b, "reads_per_launch": args.launch_reads,
                         "random_64B_gather_ceiling_GBs": gather_gbs, "probe_ends": probes},
        }
        if with_cands:
            c_avg = with_cands["classify_ms"] / max(with_cands["launches"], 1)
            c_bytes = mean_b + 8.0 * with_cands["pairs_per_read"]
            out["value_with_candidates"] = {
                "value": args.batch * args.steps * world / with_cands["dt"], "unit": "reads/s", "ms_per_step": with_cands["dt"] / args.steps * 1e3,
                "pairs_per_read": with_cands["pairs_per_read"], "algorithmic_bytes_per_read": c_bytes, "kernel_avg_ms": c_avg,
                "roofline_frac": c_bytes * args.launch_reads / (c_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "what": "the window of `value` again with -p (prn_all = 1): every read's candidate (taxid, score) pairs written to a device "
                        f"buffer of {args.cands_per_read} pairs per read, as bin/run_rl.sh:243 runs read_label; +8 B per pair in the byte count"}
        if world == 1 and not args.no_e2e and len(read_lens) == 1:
            e2e_batch = min(args.launch_reads, 2_000_000)  # the streamed boundary in batches of 2 M reads (3 pinned slots of 300 MB)
            per_step = max(1, args.batch // e2e_batch)
            out["e2e_stream"] = e2e_stream(eng, reads, e2e_batch, args.steps * per_step, args.warmup * per_step, log)
        if world == 1 and not args.no_cpu:
            with tempfile.TemporaryDirectory() as td:
                out["cpu_baseline"] = cpu_baseline(eng, reads, args.cpu_sample, k, td)
        print(json.dumps(out))
    for rs in read_sets:
        rs.free()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
