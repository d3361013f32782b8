#!/bin/bash
# Profiles of one round, on the GPU box:  bash tools/profile_round.sh r03_a [c4]
#   1. rocprofv3 --kernel-trace --stats of the default bench command           -> gpurun_out/<tag>_stats/
#   2. rocprofv3 --pmc passes (one counter group per run, no tracing beside them) of ONE C2 launch pair per
#      arithmetic mode (tools/tune.py --spp 500 --rounds 1)                        -> gpurun_out/<tag>_pmc_<mode>_<group>/
#   3. with a second argument "c4": the same two steps for the synthetic 1M-sphere scene (exact mode only: DESIGN.md 3.1),
#      kernel trace of `bench.py --config c4` and the memory-side groups (FETCH_SIZE / WRITE_SIZE / L1 + L2 hit counters)
# The summaries are made afterwards, on the CPU box, with tools/rocpd_summary.py and tools/pmc_table.py.
# Every step runs under its own time limit; the first failure ends the script (nothing else is started on the GPU).
set -u
tag=${1:-r03}
with_c4=${2:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o bench -- python3 $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err || { echo "stats pass failed"; exit 1; }
for mode in exact fast; do
  for grp in "a:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" \
             "b:SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "c:FETCH_SIZE" "d:WRITE_SIZE" "e:GRBM_GUI_ACTIVE GRBM_COUNT" \
             "f:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out/${tag}_pmc_${mode}_$name -o p -- python3 $R/tools/tune.py --spp 500 --rounds 1 "CRT_ARITH=$mode" > $out/${tag}_pmc_${mode}_$name.log 2>&1 || { echo "pmc pass $mode $name failed"; tail -3 $out/${tag}_pmc_${mode}_$name.log; exit 1; }
  done
done
if [ "$with_c4" = "c4" ]; then
  C4="--half 500 --ground 100000 --lookfrom 520 80 120 --focus 540 --width 1920 --spp 64 --rounds 1"
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_c4_stats -o bench -- python3 $R/bench.py --config c4 --steps 4 --warmup 1 --cpu-seconds 0 > $out/${tag}_c4_bench_under_rocprof.json 2> $out/${tag}_c4_stats.err || { echo "c4 stats pass failed"; exit 1; }
  for grp in "a:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM" \
             "c:FETCH_SIZE" "d:WRITE_SIZE" "e:GRBM_GUI_ACTIVE GRBM_COUNT" \
             "g:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
             "h:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    timeout -s KILL 180 rocprofv3 --pmc $ctrs --output-format csv -d $out/${tag}_pmc_c4_$name -o p -- python3 $R/tools/tune.py $C4 "CRT_ARITH=exact" > $out/${tag}_pmc_c4_$name.log 2>&1 || { echo "pmc pass c4 $name failed"; tail -3 $out/${tag}_pmc_c4_$name.log; exit 1; }
  done
fi
# the walker/owner instance (SCHED=3): kernel time, lane utilisation and the LDS / wait counters of one C2 launch
for grp in "a:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" \
           "f:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"; do
  name=${grp%%:*}; ctrs=${grp#*:}
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out/${tag}_pmc_sched3_$name -o p -- python3 $R/tools/tune.py --spp 500 --rounds 1 "CRT_SCHED=3 CRT_SHADE_NUM=6" > $out/${tag}_pmc_sched3_$name.log 2>&1 || { echo "pmc pass sched3 $name failed"; tail -3 $out/${tag}_pmc_sched3_$name.log; exit 1; }
done
echo "profiles done"
