#!/usr/bin/env python3
"""The counters of a rocprofv3 --pmc rocpd database for the BIG launches of one kernel only (default: the G1 bucket
accumulation, whose h-MSM launch fills the chip while its three other launches per proof leave most of it empty): per
counter, the mean over the launches whose value exceeds half of the largest one.  A per-kernel average over all four
launches says little about the launch that matters.
usage: rocpd_counters_big_launch.py results.db [kernel-substring]"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "k_accum_affine_g1s"
rows = db.execute("select kernel_name, counter_name, value, start from counters_collection").fetchall()
t_tab = max([r[3] for r in rows if "k_table_next" in r[0]] + [0])
by = {}
for n, c, v, s in rows:
    if s >= t_tab and pat in n:
        by.setdefault(c, []).append(v)
cs = sorted(by)
print("| kernel (big launches only) | launches | " + " | ".join(cs) + " |")
print("|---|---|" + "---|" * len(cs))
cells, k = [], 0
for c in cs:
    big = [v for v in by[c] if v > 0.5 * max(by[c])]
    k = max(k, len(big))
    cells.append("%.4g" % (sum(big) / len(big)) if big else "-")
print("| %s | %d | " % (pat, k) + " | ".join(cells) + " |")
