#!/usr/bin/env python3
"""GPU busy analysis of a rocprofv3 rocpd database: union of kernel intervals vs wall, and per-kernel share of
'attributed' time (each instant's time split equally among the kernels running then)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else -1   # ignore the first N ms; default: everything up to the last window-table build
t0 = rows[0][1] + skip * 1_000_000 if skip >= 0 else max([e for n, s, e in rows if "k_table_next" in n] + [rows[0][1]])
rows = [(re.sub(r"\(.*", "", n).replace("void ", "")[:70], s, e) for n, s, e in rows if s >= t0]
ev = []
for i, (n, s, e) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set(); last = ev[0][0]; busy = 0; attributed = {}
for t, kind, i in ev:
    if active:
        dt = t - last; busy += dt
        share = dt / len(active)
        for j in active:
            attributed[rows[j][0]] = attributed.get(rows[j][0], 0) + share
    last = t
    if kind == 1: active.add(i)
    else: active.discard(i)
wall = ev[-1][0] - ev[0][0]
print("wall %.2f ms  busy(union) %.2f ms  = %.1f%%" % (wall / 1e6, busy / 1e6, 100 * busy / wall))
for n, v in sorted(attributed.items(), key=lambda kv: -kv[1])[:22]:
    print("%6.2f%%  %8.2f ms  %s" % (100 * v / busy, v / 1e6, n))
