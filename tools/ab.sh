#!/bin/bash
# Alternating-process A/B of two builds of the library on one box (C2 geometry, exact and fast arithmetic):
#   cp cuda_ray_tracer_amd/libcrt_hip.so cuda_ray_tracer_amd/libcrt_ab_base.so   (the build to compare against)
#   ... change, make ...;  gpurun -- 'bash tools/ab.sh <log name> [rounds]'
set -e
log=gpurun_out/${1:-ab}.log; : > $log
for i in $(seq ${2:-3}); do
  for n in ${AB_LIBS:-ab_base hip}; do
    echo "== $n" >> $log
    CRT_LIB_PATH=$PWD/cuda_ray_tracer_amd/libcrt_$n.so timeout -k 10 120 python3 tools/tune.py --spp 500 --rounds 3 "CRT_BLOCK=0" "CRT_ARITH=fast" 2>&1 | grep -E "median|checksum" | sed "s/^/$n /" | tee -a $log
  done
done
