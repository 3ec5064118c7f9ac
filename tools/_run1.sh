set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tools/rows2d_copy > gpurun_out/r4_rows2d_copy.txt 2>&1
for a in "0 0" "4000 5800"; do tools/rows2d_sched $a; done > gpurun_out/r4_rows2d_sched2.txt 2>&1
python tools/ab_variants.py 2d:64 lib=sp0:0 lib=sp1:0 --rounds 5 > gpurun_out/r4_ab_spread2.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_r4_c3base -o t -- python3 tools/prof_workload.py c3 40 > gpurun_out/trace_r4_c3base.log 2>&1
cat gpurun_out/r4_rows2d_copy.txt gpurun_out/r4_rows2d_sched2.txt gpurun_out/r4_ab_spread2.txt
head -5 gpurun_out/trace_r4_c3base/*/*kernel_stats.csv
