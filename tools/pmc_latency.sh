#!/bin/bash
# Runs ON THE GPU BOX: latency / pipe-occupancy counters of the rollout kernel.  usage: tools/pmc_latency.sh <tag>
set -u
OUT=$PWD/gpurun_out/${1:-lat}; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python $PWD/bench.py --no-cpu-baseline --legs none --steps 20 --warmup 3"
cd /tmp
for grp in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- $BENCH > /dev/null 2> $OUT/pmc_$name.err
done
python $OLDPWD/tools/pmc_extract.py $OUT rollout | tee $OUT/pmc_latency.json
