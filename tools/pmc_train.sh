set -u
OUT=$PWD/gpurun_out/pmc_train; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $PWD/tools/bench_train.py"
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- $CMD > /dev/null 2> $OUT/pmc_$name.err
done
for k in dw_adam chain_kernel; do echo "== $k"; python $OLDPWD/tools/pmc_extract.py $OUT $k; done
