#!/bin/bash
# usage (on GPU box): pmc_ab.sh cfg lib...   -> icache / wait counters of the rollout kernel per library
export TMPDIR=/tmp
ROOT=$PWD
CFG=$1; shift
for lib in "$@"; do
  for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU"; do
    n=$(echo $grp | cut -c1-12)
    out=/tmp/pmc_${lib}_$n
    rm -rf $out
    (cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -- python $ROOT/tools/variant_time.py $CFG $lib > /dev/null 2> $out.err)
    python - <<PY
import csv,glob
acc={};cnt={}
for p in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        if "rollout_xdl" not in row["Kernel_Name"]: continue
        k=row["Counter_Name"]; acc[k]=acc.get(k,0)+float(row["Counter_Value"]); cnt[k]=cnt.get(k,0)+1
print("$lib", {k: round(acc[k]/cnt[k]) for k in sorted(acc)}, max(cnt.values()) if cnt else 0)
PY
  done
done
