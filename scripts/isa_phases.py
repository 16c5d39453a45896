#!/usr/bin/env python3
"""Static instruction mix of a kernel between `; MARK Lnnn` comments (asm volatile markers placed at source lines in a scratch
copy of kernels.hip, compiled with `hipcc -S --cuda-device-only`): which source region carries how many vector / scalar /
branch / LDS / memory instructions.  Static counts -- loops count once -- but the front end of the classify kernel is
straight-line code, and a region's mix shows where the compiler spends moves, spills (v_writelane / v_readlane) and waits.
usage: isa_phases.py file.s kernel-name-substring [--dump Lnnn]"""
import collections
import re
import sys

fn, want = sys.argv[1], sys.argv[2]
dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--dump" else None
lines = open(fn).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and l.split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].lstrip().startswith("s_endpgm"))
seg, order = "entry", ["entry"]
cnt = collections.defaultdict(collections.Counter)
top = collections.defaultdict(collections.Counter)
for l in lines[start:end + 1]:
    t = l.strip()
    m = re.match(r";\s*MARK (L\d+)", t)
    if m:
        seg = m.group(1)
        if seg not in order:
            order.append(seg)
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    if dump and seg == dump:
        print("   ", t)
    if op.startswith("v_"):
        kind = "VALU"
    elif op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64"):
        kind = "BR"
    elif op in ("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep"):
        kind = "misc"
    elif op.startswith("s_load") or op.startswith("s_buffer"):
        kind = "SMEM"
    elif op.startswith("s_"):
        kind = "SALU"
    elif op.startswith("ds_"):
        kind = "LDS"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        kind = "VMEM"
    else:
        kind = "other"
    cnt[seg][kind] += 1
    top[seg][op] += 1
tot = collections.Counter()
for s in order:
    c = cnt[s]
    tot.update(c)
    mv = top[s]["v_mov_b32_e32"] + top[s]["v_mov_b64_e32"] + top[s]["v_accvgpr_write_b32"] + top[s]["v_accvgpr_read_b32"]
    sp = top[s]["v_writelane_b32"] + top[s]["v_readlane_b32"] + top[s]["v_readfirstlane_b32"]
    print(f"{s:>7}: VALU {c['VALU']:5d} (mov {mv:4d}, lane-rw {sp:4d})  SALU {c['SALU']:5d}  BR {c['BR']:4d}  LDS {c['LDS']:4d}  VMEM {c['VMEM']:3d}  SMEM {c['SMEM']:3d}  wait/nop {c['misc']:4d}")
print("  total:", dict(tot))
if len(sys.argv) > 3 and sys.argv[3] == "--top":
    s = sys.argv[4]
    for op, n in top[s].most_common(40):
        print(f"   {op:28s} {n}")
