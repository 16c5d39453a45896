D=/dev/shm/clib
python scripts/cli_bench.py --db-gb 64 --reads 2000000 --repeat 3 --threads 8 --reps 1 --only "-a (calls" --keep $D > /dev/null 2>&1
ls -la $D | head
ARGS="-f $D/map32to16.txt -u $D/rank_names.txt -w $D/rank.txt -x 0 -j 30 -l 0 -b 1.0 -e $D/depth.dat -t 8 -i $D/reads_x3.fa -d $D/db.img2 -c $D/tax.dat -o $D/out -p"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04m/kt -o kt -- $GRAFT_REPO_ROOT/lmat_amd/csrc/read_label $ARGS > $GRAFT_REPO_ROOT/gpurun_out/r04m/cli_prof.out 2>&1
LMAT_CLI_TIMING=1 LMAT_DEBUG_STREAM=1 $GRAFT_REPO_ROOT/lmat_amd/csrc/read_label $ARGS > $GRAFT_REPO_ROOT/gpurun_out/r04m/cli_dbg.out 2> $GRAFT_REPO_ROOT/gpurun_out/r04m/cli_dbg.err
rm -rf $D
