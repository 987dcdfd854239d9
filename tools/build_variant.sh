#!/bin/bash
# Build an experimental variant of the DEVELOPER library next to the tree's own build (for tools/ab.sh / tools/compare_libs.py):
#   tools/build_variant.sh NAME "-DFLAG=.." [HIDS] [CTXS]   ->  cadm_amd/libcadm_hip_var_NAME.so
# The sources are copied to /tmp so the tree's objects stay those of the product build.
set -e
NAME=$1; FLAGS=$2; HIDS=${3:-200}; CTXS=${4:-"0 10"}
ROOT=$(cd $(dirname $0)/.. && pwd)
W=/tmp/cadm_var_$NAME
rm -rf $W && mkdir -p $W/cadm_amd $W/include
cp -r $ROOT/cadm_amd/csrc $W/cadm_amd/ && cp $ROOT/include/*.h $W/include/
find $W -name "*.o" -delete
make -C $W/cadm_amd/csrc -j8 dev EXTRA="$FLAGS" HIDS="$HIDS" CTXS="$CTXS" 2>&1 | grep -E "error|Error" || true
cp $W/cadm_amd/libcadm_hip_dev.so $ROOT/cadm_amd/libcadm_hip_var_$NAME.so
ls -la $ROOT/cadm_amd/libcadm_hip_var_$NAME.so
