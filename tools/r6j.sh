bash tools/r6_robust.sh 2>&1 | tail -40
