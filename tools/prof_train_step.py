"""Per-kernel breakdown of one training step from a rocprofv3 rocpd database
(rocprofv3 --kernel-trace -d DIR -o NAME -- python tools/bench_train.py  ->  DIR/NAME_results.db)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, grid_x, grid_y, grid_z, (end-start), start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "assemble" in r[0]]
step = rows[idx[-2]:idx[-1]]
print("kernels/step", len(step), "span %.1f us" % ((step[-1][6] - step[0][5]) / 1e3))
for r in step:
    name = r[0].replace("(anonymous namespace)::", "").split("(")[0][:44]
    print("%-44s grid %5d %3d %3d  %7.1f us" % (name, r[1] // 256 if r[1] >= 256 else r[1], r[2], r[3], r[4] / 1e3))
