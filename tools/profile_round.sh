#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): per-kernel rocprofv3 table + PMC passes of the bench command; summaries go to
# gpurun_out/$1/ and are then copied into profiles/ by hand.   usage: tools/profile_round.sh r2_a [bench args..]
set -u
TAG=${1:-r2}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --no-extras --steps 20 --warmup 3 $*"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python $OLDPWD/tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- $BENCH > /dev/null 2> $OUT/pmc_$name.err
done
BID=$(cd $OLDPWD && python -c "from cadm_amd import _lib; print(_lib.load().cadm_build_id().decode())")
python $OLDPWD/tools/pmc_extract.py $OUT rollout $BID > $OUT/pmc_rollout.json
cat $OUT/pmc_rollout.json
