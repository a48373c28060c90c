#!/usr/bin/env python3
"""Per-kernel, per-launch averages of every counter in a rocprofv3 --pmc rocpd database (steady-state proofs only:
after the last window-table build).  usage: rocpd_counters.py results.db [out.md]"""
import re, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, value, start from counters_collection").fetchall()
t_tab = max([r[3] for r in rows if "k_table_next" in r[0]] + [0])
agg = {}
for n, c, v, s in rows:
    if s < t_tab:
        continue
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    n = n.replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::").replace("rocprim::ROCPRIM_400200_NS::", "rp::")
    n = re.sub(r"rp::trampoline_kernel<rp::wrapped_(\w+?)_config<.*", r"rocprim \1", n)[:64]
    a = agg.setdefault(n, {}).setdefault(c, [0.0, 0])
    a[0] += v; a[1] += 1
ctrs = sorted({c for k in agg.values() for c in k})
lines = ["| kernel | launches | " + " | ".join(ctrs) + " |", "|---|---|" + "---|" * len(ctrs)]
for n, cs in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1].values())):
    k = max(x[1] for x in cs.values())
    lines.append("| %s | %d | " % (n, k) + " | ".join("%.4g" % (cs[c][0] / cs[c][1]) if c in cs else "-" for c in ctrs) + " |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
