#!/bin/bash
# scripts/build_variant.sh NAME SOURCE [extra hipcc flags]: liblmat_hip.so with kernels.hip replaced by SOURCE -> lmat_amd/variants/NAME.so
# (same-box A/B experiments: scripts/ab_libs.sh, scripts/pmc_ablate.sh with LMAT_LIB)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=$1; SRC=$2; shift 2
O=$ROOT/lmat_amd/csrc
mkdir -p $ROOT/lmat_amd/variants /tmp/kv
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off --offload-arch=gfx950 -I$O -Wno-unused-result -Wno-unused-function -Wno-unused-value -Wno-pass-failed "$@" -x hip -c $SRC -o /tmp/kv/$N.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/lmat_amd/variants/$N.so /tmp/kv/$N.o $O/lmat_api.o $O/taxonomy.o $O/dbbuild.o $O/nullmodel.o $O/collective.o -lz -lpthread -ldl && echo built $N
