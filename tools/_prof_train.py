import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, grid_x, grid_y, grid_z, (end-start), start, end from kernels order by start").fetchall()
n = len(rows)
# last step: find period by locating assemble kernels
idx = [i for i, r in enumerate(rows) if 'assemble' in r[0]]
a, b = idx[-2], idx[-1]
step = rows[a:b]
print("kernels/step", len(step), "span %.1f us" % ((step[-1][6]-step[0][5])/1e3), "sum dur %.1f us" % (sum(r[4] for r in step)/1e3))
agg = collections.OrderedDict()
for r in step:
    k = r[0].split('(')[0][-40:]
    d = agg.setdefault(k, [0, 0]); d[0] += 1; d[1] += r[4]
for k, (c, t) in agg.items(): print("%-42s n=%3d total %.1f us avg %.2f" % (k, c, t/1e3, t/c/1e3))
for r in step:
    if 'gemm' in r[0]: print(r[1]//256, r[2], r[3], "%.1f" % (r[4]/1e3), end=" | ")
print()
