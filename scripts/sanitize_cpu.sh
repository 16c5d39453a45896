#!/bin/bash
# CPU-side sanitizers (sanitizers row): the oracle, the host objects of liblmat_hip.so and the GPU-free tools built
# with AddressSanitizer + UndefinedBehaviorSanitizer (ROCm's clang and its runtimes), then the whole `-m "not gpu"` suite run
# against them.  GPU AddressSanitizer is not available on this pool: device code is not instrumented.
# Writes profiles/r04_sanitizers.txt.  The tools are swapped in place for the run and rebuilt afterwards.
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CL=/opt/rocm/lib/llvm/bin/clang++
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -shared-libasan"
OUT=/tmp/lmat_san
LOG=$ROOT/profiles/r04_sanitizers.txt
rm -rf $OUT && mkdir -p $OUT
cd $ROOT/oracle
$CL -std=gnu++17 -O1 -g -ffp-contract=off $SAN -shared -fPIC -o $OUT/liblmat_oracle.so lmat_oracle_capi.cpp -lpthread -lz || exit 1
cd $ROOT/lmat_amd/csrc
for f in lmat_api taxonomy dbbuild nullmodel collective; do
  $CL -std=c++17 -O1 -g -fPIC -ffp-contract=off $SAN -Wno-unused-value -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $f.cpp -o $OUT/$f.o || exit 1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $SAN -o $OUT/liblmat_hip.so kernels.o $OUT/lmat_api.o $OUT/taxonomy.o $OUT/dbbuild.o $OUT/nullmodel.o $OUT/collective.o -lz -lpthread -ldl || exit 1
mkdir -p $OUT/keep && cp make_db_image content_summ fs_rollup fmt_check gene_label $OUT/keep/
restore() { cp $OUT/keep/* $ROOT/lmat_amd/csrc/; }
trap restore EXIT
$CL -std=c++17 -O1 -g $SAN -o make_db_image make_db_image_main.cpp dbbuild.cpp || exit 1
$CL -std=c++17 -O1 -g $SAN -o content_summ content_summ_main.cpp -lpthread || exit 1
$CL -std=c++17 -O1 -g $SAN -o fs_rollup fs_rollup_main.cpp || exit 1
$CL -std=c++17 -O1 -g $SAN -o fmt_check fmt_check.cpp || exit 1
$CL -std=c++17 -O1 -g -ffp-contract=off $SAN -o gene_label gene_label_main.cpp -L$OUT -llmat_hip -Wl,-rpath,$OUT -Wl,-rpath-link,/opt/rocm/lib -lz || exit 1
cd $ROOT
{
  echo "# CPU-side sanitizer run: ASan + UBSan (clang $($CL --version | head -1 | sed 's/.*version //'))"
  echo "# instrumented: oracle/lmat_oracle_capi.cpp (+ lmat_oracle.hpp, gene_oracle.hpp); lmat_api.cpp taxonomy.cpp dbbuild.cpp nullmodel.cpp collective.cpp"
  echo "#   (host objects of liblmat_hip.so; kernels.o as built); make_db_image content_summ fs_rollup fmt_check gene_label"
  echo "# command: LD_PRELOAD=libclang_rt.asan LMAT_LIB=... LMAT_ORACLE_LIB=... python -m pytest tests -m 'not gpu' -q   ($(date -u +%F))"
} > $LOG
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  LMAT_LIB=$OUT/liblmat_hip.so LMAT_ORACLE_LIB=$OUT/liblmat_oracle.so \
  timeout 3000 python -m pytest tests -m "not gpu" -q -x -p no:cacheprovider 2>&1 | tail -25 | tee -a $LOG
rc=${PIPESTATUS[0]}
echo "# exit code $rc; sanitizer reports in the output above: $(grep -c 'ERROR: AddressSanitizer\|runtime error:' $LOG)" >> $LOG
exit $rc
