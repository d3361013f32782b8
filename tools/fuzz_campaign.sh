#!/bin/bash
# The long parity campaigns of a round, on the GPU box:  bash tools/fuzz_campaign.sh r03
# One summary line per campaign is appended to gpurun_out/<tag>_fuzz_campaigns.txt (copied to profiles/ afterwards).
set -u
tag=${1:-r03}
cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_fuzz_campaigns.txt
id=$(python3 -c "import cuda_ray_tracer_amd as c; print(c.build_info()['build_id'])")
run() {  # name, environment assignments, pytest -k expression
  local name=$1 envs=$2 expr=$3 t0=$(date +%s)
  local line
  line=$(env $envs timeout -k 10 900 python3 -u -m pytest tests/test_gpu_parity.py -q -k "$expr" --timeout 600 --timeout-method thread 2>&1 | tail -1)
  local rc=$?
  echo "$name: build $id, env [$envs], pytest -k '$expr': $line, wall $(( $(date +%s) - t0 )) s" | tee -a $out
  case "$line" in *failed*|*error*) exit 1;; esac
}
: > $out
run "scene/camera/depth fuzz, 12000 seeded cases (seeds 1000..12999), frame + ray count + traversal counters vs oracle" "CRT_FUZZ_CASES=12000" "fuzz_scene" || exit 1
run "the same, first 600 cases, walker/owner instance" "CRT_FUZZ_CASES=600 CRT_FUZZ_ENV=CRT_SCHED=3" "fuzz_scene" || exit 1
run "the same, first 600 cases, scene read from HBM (no LDS image)" "CRT_FUZZ_CASES=600 CRT_FUZZ_ENV=CRT_LDS_LIMIT=0" "fuzz_scene" || exit 1
run "GPU BVH builder vs host replay of the reference's recursion, 14 fixed + 200 random sizes in 1..20000, tie-heavy and random scenes" "CRT_BVH_FUZZ_SIZES=200" "bvh_builder or bvh_readback" || exit 1
echo "campaigns done"
