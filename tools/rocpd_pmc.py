#!/usr/bin/env python3
"""Per-kernel sums of a PMC counter from a rocprofv3 rocpd database, per proof.
usage: rocpd_pmc.py results.db COUNTER [out.md]   (proofs are counted by k_w_to29 dispatches)"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
ctr = sys.argv[2]
rows = cur.execute("select kernel_name, counter_name, value, start, dispatch_id from counters_collection").fetchall()
# steady state only: after the last window-table build (load or the one-time re-tune following the first proof)
t_tab = max([r[3] for r in rows if "k_table_next" in r[0]] + [0])
t_first = min(r[3] for r in rows if "k_w_to29" in r[0] and r[3] > t_tab)
nproofs = len(set(r[4] for r in rows if "k_w_to29" in r[0] and r[1] == ctr and r[3] >= t_first))
agg = {}
for n, c, v, s, d in rows:
    if s < t_first or c != ctr:
        continue
    n = re.sub(r"\(.*", "", n).replace("void ", "")[:80]
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += v
tot = sum(a[1] for a in agg.values())
lines = ["%s per proof (%d steady-state proofs, after the one-time window re-tune)" % (ctr, nproofs), "",
         "| kernel | launches/proof | %s per proof (%s) | %% |" % (ctr, "MB: the counter is in KB" if ctr.endswith("_SIZE") else "M"),
         "|---|---|---|---|"]
unit = 1e3 if ctr.endswith("_SIZE") else 1e6
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if a[1] / tot < 0.0005:
        continue
    lines.append("| %s | %.1f | %.1f | %.1f |" % (n, a[0] / nproofs, a[1] / nproofs / unit, 100 * a[1] / tot))
lines += ["", "total: %.1f %s per proof" % (tot / nproofs / unit, "MB" if ctr.endswith("_SIZE") else "M")]
if ctr == "SQ_INSTS_VALU":
    lines.append("at the measured issue ceiling of 578 G wave-instructions/s (tools/ubench: 168 G Fq products/s x 220 instr / 64 lanes): %.2f ms per proof" % (tot / nproofs / 578e9 * 1e3))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
