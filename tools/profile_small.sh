#!/bin/bash
# Kernel traces of plans that do not fill the chip, the library's choice against the large-batch split with streaming accesses
# (variant 536870912): tools/profile_small.sh TAG [WORKLOAD ...]  -> gpurun_out/trace_TAG_<workload>/ ; fold with tools/steady_stats.py
set -eo pipefail
TAG=$1
shift
WLS=("$@")
[ ${#WLS[@]} -eq 0 ] && WLS=("n2^20:1" "n2^20:1:536870912" "n2^20:16" "n2^20:16:536870912" "n2^24:1" "n2^24:1:536870912" "n2^18:1" "n2^18:1:536870912")
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
python3 -c 'import __graft_entry__ as g; g.build()'
for WL in "${WLS[@]}"; do
  D=$OUT/trace_${TAG}_$(echo $WL | tr ':^' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python3 tools/prof_workload.py $WL 200 400 > $D.log 2>&1
  python3 tools/steady_stats.py "$D/*kernel_trace.csv" --warmup 400 --reps 200 --out $D.csv > /dev/null
  echo "== $WL"; cat $D.csv
done
