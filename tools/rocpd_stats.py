#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database into a per-kernel table (like `--stats` CSV output).
usage: rocpd_stats.py results.db [out.md]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, (end - start), start from kernels" % name_col).fetchall()
# load time (key generation, window tables, the one-time re-tune) ends with the last window-table kernel: what comes after is
# the prove path.  A profile of bench.py as it is spends most of its traced time before that point (round 4: k_ec_stage 54 %
# of it), so the two are reported apart - the prove-path table first.
t_load = max([s for n, d, s in rows if "k_table_next" in n] + [0])
steady = [(n, d) for n, d, s in rows if s > t_load]
load = [(n, d) for n, d, s in rows if s <= t_load]
agg = {}
for n, d in (steady if steady else [(n, d) for n, d, s in rows]):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void ", "", n)
    a = agg.setdefault(n, [0, 0, 10**18, 0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (n[:110], a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
if load and steady:
    lagg = {}
    for n, d in load:
        n = re.sub(r"^void ", "", re.sub(r"\(.*", "", n))
        a = lagg.setdefault(n, [0, 0])
        a[0] += 1; a[1] += d
    ltot = sum(a[1] for a in lagg.values())
    lines = ["prove path (after the last window-table build): %.1f ms of kernels; load time before it: %.1f ms (second table)" % (tot / 1e6, ltot / 1e6), ""] + lines
    lines += ["", "| load-time kernel | calls | total ms | % of load time |", "|---|---|---|---|"]
    for n, a in sorted(lagg.items(), key=lambda kv: -kv[1][1])[:12]:
        lines.append("| %s | %d | %.3f | %.1f |" % (n[:110], a[0], a[1] / 1e6, 100.0 * a[1] / ltot))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
