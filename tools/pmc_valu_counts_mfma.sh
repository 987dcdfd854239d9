#!/bin/bash
# Does SQ_INSTS_VALU count MFMA instructions?  (reading of the "VALU per MFMA" figures of profiles/r6_valu_ledger.md)
# tools/micro/mfma_k16_rate.hip: three kernels of 160 000 MFMAs per wave and a handful of other VALU instructions.
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_valu
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_k16_rate.hip -o /tmp/mfma_k16_rate || exit 1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $OUT/pmc_a -- /tmp/mfma_k16_rate > $OUT/run.txt 2> $OUT/pmc_a.err
cd $OLDPWD
python - <<EOF
import csv, glob
for p in glob.glob("$OUT/pmc_a/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        print(row.get("Kernel_Name")[:60], row.get("Counter_Name"), row.get("Counter_Value"))
EOF
