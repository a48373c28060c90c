#!/usr/bin/env python3
"""Per-kernel VALU efficiency: stand-alone duration (kernel trace taken with --inflight 1 --mode throughput)
against the kernel's own instruction floor (SQ_INSTS_VALU from a --pmc pass of the same command, divided by the
measured issue ceiling of 578 G wave-instructions/s).
usage: rocpd_efficiency.py trace.db pmc.db [out.md]"""
import re, sqlite3, sys

PEAK = 578e9


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    n = n.replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::").replace("rocprim::ROCPRIM_400200_NS::", "rp::")
    n = re.sub(r"rp::trampoline_kernel<rp::wrapped_(\w+?)_config<.*", r"rocprim \1", n)
    return n[:64]


def steady(rows, name_i, start_i):
    t_tab = max([r[start_i] for r in rows if "k_table_next" in r[name_i]] + [0])
    firsts = sorted(r[start_i] for r in rows if "k_w_to29" in r[name_i] and r[start_i] > t_tab)
    return firsts[0], len(firsts)


tr = sqlite3.connect(sys.argv[1]).execute("select name, start, end from kernels").fetchall()
t0, n_tr = steady(tr, 0, 1)
dur = {}
for n, s, e in tr:
    if s >= t0:
        a = dur.setdefault(short(n), [0, 0])
        a[0] += 1; a[1] += e - s
pm = sqlite3.connect(sys.argv[2]).execute(
    "select kernel_name, start, value from counters_collection where counter_name='SQ_INSTS_VALU'").fetchall()
p0, n_pm = steady(pm, 0, 1)
valu = {}
for n, s, v in pm:
    if s >= p0:
        valu[short(n)] = valu.get(short(n), 0) + v
lines = ["stand-alone kernel time per proof vs VALU instruction floor (%d traced proofs, %d counted proofs)" % (n_tr, n_pm), "",
         "| kernel | launches/proof | us/proof stand-alone | VALU M wave-instr/proof | floor us | efficiency |", "|---|---|---|---|---|---|"]
tot_d = tot_f = 0
for n, a in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    d = a[1] / n_tr / 1e3
    v = valu.get(n, 0) / n_pm
    f = v / PEAK * 1e6
    tot_d += d; tot_f += f
    if d < 5:
        continue
    lines.append("| %s | %.1f | %.1f | %.1f | %.1f | %.0f%% |" % (n, a[0] / n_tr, d, v / 1e6, f, 100 * f / d if d else 0))
lines += ["", "total: %.2f ms stand-alone per proof, VALU floor %.2f ms (%.0f%%)" % (tot_d / 1e3, tot_f / 1e3, 100 * tot_f / tot_d)]
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
