#!/bin/bash
# gpurun_out/r6_* (tools/profile_r6.sh, tools/r6_robust.sh) -> profiles/r6_*
set -e
cd $(dirname $0)/..
G=gpurun_out
cp $G/r6_cfg2/kernel_stats.md profiles/r6_cfg2_kernel_stats.md; cp $G/r6_cfg2/bench_under_rocprof.json profiles/r6_cfg2_bench_under_rocprof.json; cp $G/r6_cfg2/pmc_rollout.json profiles/r6_pmc_cfg2.json
cp $G/r6_cfg3/kernel_stats.md profiles/r6_cfg3_kernel_stats.md; cp $G/r6_cfg3/bench_under_rocprof.json profiles/r6_cfg3_bench_under_rocprof.json; cp $G/r6_cfg3/pmc_rollout.json profiles/r6_pmc_cfg3.json
cp $G/r6_cfg3/pmc_wave_tile.json profiles/r6_pmc_cfg3_wave_tile.json; cp $G/r6_cfg3/pmc_two_tile.json profiles/r6_pmc_cfg3_two_tile.json
for B in 256 4096; do
  cp $G/r6_train_$B/kernel_stats.md profiles/r6_train_B${B}_kernel_stats.md
  python - <<PY
import json
a=json.load(open("$G/r6_train_$B/pmc_chain.json")); b=json.load(open("$G/r6_train_$B/pmc_dw_adam.json"))
json.dump({"batch": $B, "chain_kernel": a, "dw_adam_kernel": b, "bench": [l for l in open("$G/r6_train_$B/bench.txt").read().strip().splitlines() if "amdgpu.ids" not in l]}, open("profiles/r6_train_pmc_B$B.json","w"), indent=1)
PY
done
cp $G/r6_context/kernel_stats.md profiles/r6_context_batched_kernel_stats.md; tail -2 $G/r6_context/bench.txt > profiles/r6_context_batched_bench.txt
cp $G/r6_final/bench.json profiles/r6_bench.json
