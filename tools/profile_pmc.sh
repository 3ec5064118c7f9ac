#!/bin/bash
# Kernel trace + PMC counters of one workload, each counter set in its OWN rocprofv3 pass (MI355X_MICROARCH.md,
# "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE cannot share a pass; never --pmc together with a trace domain).
#   tools/profile_pmc.sh TAG WORKLOAD [reps [warmup]]      e.g.  tools/profile_pmc.sh r4_c3 c3 10 120
# Every pass runs `warmup` untimed executions in front of the `reps` counted ones (clock ramp; tools/prof_workload.py).
# Output: gpurun_out/pmc_TAG_<set>/ and gpurun_out/trace_TAG/ ; fold with
#   python tools/steady_stats.py "gpurun_out/trace_TAG/*kernel_trace.csv" --warmup WARM --reps 40 --out profiles/TAG_kernel_stats.csv
#   python tools/summarize_pmc.py TAG KERNEL ... --skip-frac WARM/(WARM+REPS) --trace profiles/TAG_kernel_stats.csv
set -eo pipefail
TAG=$1; WL=$2; REPS=${3:-10}; WARM=${4:-120}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
# build BEFORE the first profiler line: under rocprofv3 (with --pmc its preloaded library has initialised the GPU) nothing may
# start hipcc / make any more (that would be a wrapper hop after GPU initialisation, which this pool forbids)
python3 -c 'import __graft_entry__ as g; g.build()'
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$TAG -o t -- python3 tools/prof_workload.py $WL 40 $WARM > $OUT/trace_$TAG.log 2>&1
echo "trace $TAG done"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_${TAG}_$i -o p -- python3 tools/prof_workload.py $WL $REPS $WARM > $OUT/pmc_${TAG}_$i.log 2>&1
  echo "pmc $TAG set $i done"
done
