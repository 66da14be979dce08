"""Summarise a rocprofv3 sqlite trace: per-kernel totals, and per-dispatch durations of selected kernels."""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall() if 'grid_x' in cols else None
if rows is None:
    print(cols); sys.exit()
import collections
tot = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, g, w in rows:
    tot[n][0] += 1; tot[n][1] += (e - s) / 1e3
T = sum(v[1] for v in tot.values())
span = (rows[-1][2] - rows[0][1]) / 1e3
print(f"total kernel time {T/1e3:.1f} ms, span {span/1e3:.1f} ms")
for n, (k, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{us/1e3:9.2f} ms {100*us/T:5.1f}% {k:6d} calls {us/k:9.1f} us avg  {n[:110]}")
if len(sys.argv) > 2:
    pat = sys.argv[2]
    sel = [(n, (e - s) / 1e3, g) for n, s, e, g, w in rows if re.search(pat, n)]
    for n, us, g in sel[-int(sys.argv[3]) if len(sys.argv) > 3 else -40:]:
        print(f"{us:9.1f} us grid {g:8d} {n[:90]}")
