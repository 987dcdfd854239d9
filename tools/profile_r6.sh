#!/bin/bash
# Round-6 evidence run ON THE GPU BOX (via gpurun): tools/profile_r6.sh
#   cfg2 / cfg3: per-kernel rocprofv3 table + PMC passes of the bench command (tools/profile_round.sh), the wave-tile and the two-tile kernel of cfg3 apart;
#   training step at B = 256 and B = 4096: kernel table + PMC; batched context encoder: kernel table.
set -u
R=$PWD
export TMPDIR=/tmp
tools/profile_round.sh r6_cfg2 > /dev/null 2>&1
tools/profile_round.sh r6_cfg3 --config cfg3 > /dev/null 2>&1
BID=$(python -c "from cadm_amd import _lib; print(_lib.load().cadm_build_id().decode())")
python tools/pmc_extract.py gpurun_out/r6_cfg3 rollout_wt_kernel $BID > gpurun_out/r6_cfg3/pmc_wave_tile.json
python tools/pmc_extract.py gpurun_out/r6_cfg3 rollout_xdl_kernel $BID > gpurun_out/r6_cfg3/pmc_two_tile.json
for B in 256 4096; do
  OUT=$R/gpurun_out/r6_train_$B; mkdir -p $OUT
  CMD="python $R/tools/bench_train.py $B"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -- $CMD > $OUT/bench.txt 2> $OUT/trace.err)
  DB=$(find $OUT/trace -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-40)
    (cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- $CMD > /dev/null 2> $OUT/pmc_$name.err)
  done
  python tools/pmc_extract.py $OUT chain_kernel $BID > $OUT/pmc_chain.json
  python tools/pmc_extract.py $OUT dw_adam $BID > $OUT/pmc_dw_adam.json
  find $OUT -name "*.db" -delete; rm -rf $OUT/pmc_*/ $OUT/trace
done
OUT=$R/gpurun_out/r6_context; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -- python -c "
import sys; sys.path.insert(0, '$R'); import bench
print(bench.context_bench('cuda:0', m=2048)); print(bench.context_bench('cuda:0', m=8192))" > $OUT/bench.txt 2> $OUT/trace.err)
DB=$(find $OUT/trace -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null
rm -rf $OUT/trace
for T in r6_cfg2 r6_cfg3; do find gpurun_out/$T -name "*.db" -delete; rm -rf gpurun_out/$T/pmc_*/ gpurun_out/$T/trace; done
ls -la gpurun_out/r6_cfg2 gpurun_out/r6_cfg3 gpurun_out/r6_train_4096 gpurun_out/r6_context
cat gpurun_out/r6_cfg2/pmc_rollout.json | head -30
