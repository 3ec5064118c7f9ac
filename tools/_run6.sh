cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r4 && bash tools/profile_pmc.sh r4_c2 c2 10 120 && bash tools/profile_pmc.sh r4_c3 c3 10 120 && bash tools/profile_pmc.sh r4_c2ti "n1048576:1024" 10 120
ls gpurun_out | grep r4 | head -50
