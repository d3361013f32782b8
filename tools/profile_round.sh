#!/bin/bash
# Profiles of one round, on the GPU box:  bash tools/profile_round.sh r02_a
#   1. rocprofv3 --kernel-trace --stats of the default bench command           -> gpurun_out/<tag>_stats/
#   2. rocprofv3 --pmc passes (one counter group per run, no tracing beside them) of ONE C2 launch pair per
#      arithmetic mode (tools/tune.py --spp 500 --rounds 1)                        -> gpurun_out/<tag>_pmc_<mode>_<group>/
# The summaries are made afterwards, on the CPU box, with tools/rocpd_summary.py and tools/pmc_table.py.
set -u
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o bench -- python3 $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err || { echo "stats pass failed"; exit 1; }
for mode in exact fast; do
  for grp in "a:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" \
             "b:SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "c:FETCH_SIZE" "d:WRITE_SIZE" "e:GRBM_GUI_ACTIVE GRBM_COUNT"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out/${tag}_pmc_${mode}_$name -o p -- python3 $R/tools/tune.py --spp 500 --rounds 1 "CRT_ARITH=$mode" > $out/${tag}_pmc_${mode}_$name.log 2>&1 || { echo "pmc pass $mode $name failed"; tail -3 $out/${tag}_pmc_${mode}_$name.log; exit 1; }
  done
done
echo "profiles done"
