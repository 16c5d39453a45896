#!/bin/bash
# scripts/isa_lines.sh FROM TO [kernel-substring]: the headline kernel's instructions whose debug line lies in kernels.hip:[FROM, TO]
# (compiled with -gline-tables-only into /tmp/isa; pass REBUILD=1 after editing kernels.hip)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
K=${3:-classify_kernelILi160ELi64ELi256ELb0ELb0ELb1ELb0}
mkdir -p /tmp/isa
if [ -n "$REBUILD" ] || [ ! -f /tmp/isa/kg.s ] || [ $ROOT/lmat_amd/csrc/kernels.hip -nt /tmp/isa/kg.s ]; then
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off --offload-arch=gfx950 -gline-tables-only -S --cuda-device-only -o /tmp/isa/kg.s $ROOT/lmat_amd/csrc/kernels.hip 2>/dev/null
fi
python3 - "$1" "$2" "$K" <<'PY'
import re, sys
a, b, K = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
lines = open('/tmp/isa/kg.s').read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and K in l and l.split(';')[0].rstrip().endswith(':'))
cur = None; on = False; nv = ns = nb = 0
for l in lines[start:]:
    t = l.strip()
    m = re.match(r'\.loc\s+\d+\s+(\d+).*;\s*(\S+):\d+:\d+', t)
    if m:
        cur = (0 if m.group(2).endswith('kernels.hip') else 1, int(m.group(1)))
        now = cur[0] == 0 and a <= cur[1] <= b
        if now and not on: print(f'--- line {cur[1]}')
        on = now
        if on: print(f'                                        ; :{cur[1]}')
        continue
    if t.startswith('s_endpgm'): break
    if not on or not t or t[0] in ';.': continue
    print(l)
    op = t.split()[0]
    if op.startswith('v_'): nv += 1
    elif op.startswith('s_cbranch') or op == 's_branch': nb += 1
    elif op.startswith('s_') and op not in ('s_waitcnt', 's_nop'): ns += 1
print(f'; static: VALU {nv} SALU {ns} BRANCH {nb}')
PY
