#!/bin/bash
# gpurun_out/r5_* (tools/profile_r5.sh, tools/r5_robust.sh) -> profiles/r5_*
set -e
cd $(dirname $0)/..
G=gpurun_out
cp $G/r5_cfg2/kernel_stats.md profiles/r5_cfg2_kernel_stats.md; cp $G/r5_cfg2/bench_under_rocprof.json profiles/r5_cfg2_bench_under_rocprof.json; cp $G/r5_cfg2/pmc_rollout.json profiles/r5_pmc_cfg2.json
cp $G/r5_cfg3/kernel_stats.md profiles/r5_cfg3_kernel_stats.md; cp $G/r5_cfg3/bench_under_rocprof.json profiles/r5_cfg3_bench_under_rocprof.json; cp $G/r5_cfg3/pmc_rollout.json profiles/r5_pmc_cfg3.json
cp $G/r5_cfg3/pmc_wave_tile.json profiles/r5_pmc_cfg3_wave_tile.json; cp $G/r5_cfg3/pmc_two_tile.json profiles/r5_pmc_cfg3_two_tile.json
for B in 256 4096; do
  cp $G/r5_train_$B/kernel_stats.md profiles/r5_train_B${B}_kernel_stats.md
  python - <<PY
import json
a=json.load(open("$G/r5_train_$B/pmc_chain.json")); b=json.load(open("$G/r5_train_$B/pmc_dw_adam.json"))
json.dump({"batch": $B, "chain_kernel": a, "dw_adam_kernel": b, "bench": [l for l in open("$G/r5_train_$B/bench.txt").read().strip().splitlines() if "amdgpu.ids" not in l]}, open("profiles/r5_train_pmc_B$B.json","w"), indent=1)
PY
done
cp $G/r5_context/kernel_stats.md profiles/r5_context_batched_kernel_stats.md; tail -2 $G/r5_context/bench.txt > profiles/r5_context_batched_bench.txt
cp $G/r5_final/bench.json profiles/r5_bench.json
