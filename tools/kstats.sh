#!/bin/bash
# Per-kernel time table of a bench.py run under rocprofv3: tools/kstats.sh [bench args]   (prints the bench line's headline and the top kernels)
export TMPDIR=/tmp
R=$(cd $(dirname $0)/.. && pwd)
OUT=/tmp/kstats_$$
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py --legs none --no-cpu-baseline --no-extras "$@" > $OUT.log 2>&1
python - $OUT $OUT.log <<'PY'
import csv, glob, json, sys
out, log = sys.argv[1], sys.argv[2]
for line in open(log):
    line = line.strip()
    if line.startswith("{") and '"metric"' in line:
        d = json.loads(line)
        print("bench: %.1f M row-steps/s, %.4f ms per get_action, rollout %.1f us per launch (events)" % (d["value"] / 1e6, d["ms_per_step"], 1e3 * d["roofline"]["avg_launch_ms"]))
f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:10]:
    print("%-70s calls %6s  avg %9.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
