#!/bin/bash
# rocprofv3 evidence for the headline benchmark, from the SAME command the driver runs (python3 bench.py):
#   kernel trace + stats of a default run, then PMC counters in separate passes (FETCH_SIZE and WRITE_SIZE cannot share
#   one; never --pmc together with a trace domain).   usage: tools/profile_bench.sh TAG
# Output under gpurun_out/; fold with (steady state only: the trace run issues 200 cold + 100 ramp + 20 warm-up launches in front
# of its 200 timed + 20 single ones, a PMC pass 3 + 100 + 1 in front of 3 + 20):
#   python tools/steady_stats.py "gpurun_out/trace_TAG_bench/*kernel_trace.csv" --only fft4096_kernel --skip 320 --out profiles/TAG_bench_kernel_stats.csv
#   python tools/summarize_pmc.py TAG fft4096_kernel --dirs "gpurun_out/pmc_TAG_bench_*" --skip-frac 0.82 \
#          --trace profiles/TAG_bench_kernel_stats.csv --alg-bytes 2147483648
set -eo pipefail
TAG=$1
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
# build BEFORE the first profiler line: under rocprofv3 (with --pmc its preloaded library has initialised the GPU) nothing may
# start hipcc / make any more (that would be a wrapper hop after GPU initialisation, which this pool forbids)
python3 -c 'import __graft_entry__ as g; g.build()'
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_${TAG}_bench -o t -- python3 bench.py --no-cpu-baseline --no-other-configs > $OUT/trace_${TAG}_bench.log 2>&1
echo "trace done"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_${TAG}_bench_$i -o p -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 > $OUT/pmc_${TAG}_bench_$i.log 2>&1
  echo "pmc set $i done"
done
