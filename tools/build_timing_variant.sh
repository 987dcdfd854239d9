#!/bin/bash
# Phase-timing build (CADM_PHASE_TIMING) of an experimental variant, for tools/phase_timing.py / tools/sweep_trace.py:
#   tools/build_timing_variant.sh NAME "-DFLAG=.." [HIDS] [CTXS]   ->  cadm_amd/libcadm_hip_timing_NAME.so
set -e
NAME=$1; FLAGS=$2; HIDS=${3:-200}; CTXS=${4:-"0 10"}
ROOT=$(cd $(dirname $0)/.. && pwd)
W=/tmp/cadm_tvar_$NAME
rm -rf $W && mkdir -p $W/cadm_amd $W/include
cp -r $ROOT/cadm_amd/csrc $W/cadm_amd/ && cp $ROOT/include/*.h $W/include/
find $W -name "*.o" -delete
make -C $W/cadm_amd/csrc -j8 timing EXTRA="$FLAGS" HIDS="$HIDS" CTXS="$CTXS" 2>&1 | grep -E "error|Error" || true
cp $W/cadm_amd/libcadm_hip_timing.so $ROOT/cadm_amd/libcadm_hip_timing_$NAME.so
ls -la $ROOT/cadm_amd/libcadm_hip_timing_$NAME.so
