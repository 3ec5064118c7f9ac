#!/bin/bash
# Round 5 evidence runs: kernel trace + PMC passes of the three-pass plans at 2^24 / 2^26 with the copy
# ceilings of their row pitches, and of the N = 256 ... 2048 single-pass kernels.  tools/profile_round5.sh [part]   (part: big | small)
set -eo pipefail
cd "$(dirname "$0")/.."
PART=${1:-all}
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c 'import __graft_entry__ as g; g.build()'
if [ "$PART" = all ] || [ "$PART" = big ]; then
  tools/profile_pmc.sh r5_n2^24 "n2^24:64" 10 120
  tools/profile_pmc.sh r5_n2^26x1 "n2^26:1" 10 120
  tools/profile_pmc.sh r5_n2^26x16 "n2^26:16" 10 60
  for P in 65536 32768 262144 524288; do ./tools/conc_bench $P; done > gpurun_out/r5_conc_pitches.txt 2>&1
fi
if [ "$PART" = all ] || [ "$PART" = small ]; then
  tools/profile_pmc.sh r5_n256 "n256:1048576" 10 120
  tools/profile_pmc.sh r5_n1024 "n1024:262144" 10 120
  for WL in "n512:524288" "n2048:131072"; do
    D=gpurun_out/trace_r5_$(echo $WL | tr ':' '_')
    rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python3 tools/prof_workload.py $WL 40 120 > $D.log 2>&1
    echo "trace $WL done"
  done
fi
