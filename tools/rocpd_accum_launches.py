#!/usr/bin/env python3
"""k_accum_affine launch by launch: stand-alone duration (kernel trace taken with --inflight 1 --mode throughput) against
the launch's own VALU instruction floor (SQ_INSTS_VALU of the same launch from a --pmc pass of the same command / 578 G
wave-instructions per second), for the accumulation launches between one k_w_to29 and the next: the witness map sits
between a proof's assignment-driven MSMs and its h MSM (one-stream order l, a, b1, b2 [G2], witness map, h), so such a
window holds h of one proof and then l, a, b1, b2 of the next.  Shows how much of the kernel's blended "efficiency" is the chip-filling h launch and how much the
small MSMs, whose launches leave most of the chip empty when they run alone.
usage: rocpd_accum_launches.py trace.db pmc.db [out.md]"""
import re, sqlite3, sys

PEAK = 578e9


def steady(rows, name_i, start_i):
    t_tab = max([r[start_i] for r in rows if "k_table_next" in r[name_i]] + [0])
    firsts = sorted(r[start_i] for r in rows if "k_w_to29" in r[name_i] and r[start_i] > t_tab)
    return firsts


tr = sqlite3.connect(sys.argv[1]).execute("select name, start, end from kernels order by start").fetchall()
pm = sqlite3.connect(sys.argv[2]).execute(
    "select kernel_name, start, value from counters_collection where counter_name='SQ_INSTS_VALU' order by start").fetchall()


def per_proof(rows, firsts, val):
    """{position in proof: [values]} for the accumulation launches, proofs delimited by the k_w_to29 launches"""
    out = {}
    bounds = firsts + [float("inf")]
    for p in range(len(firsts)):
        lo, hi = bounds[p], bounds[p + 1]
        acc = [r for r in rows if lo <= r[1] < hi and "k_accum_affine" in r[0]]
        for i, r in enumerate(acc):
            kind = "G2" if ("Fq2_29" in r[0] or "_g2" in r[0]) else "G1"
            out.setdefault((i, kind), []).append(val(r))
    return out


ft, fp = steady(tr, 0, 1), steady(pm, 0, 1)
dur = per_proof(tr, ft, lambda r: (r[2] - r[1]) / 1e3)
ins = per_proof(pm, fp, lambda r: r[2])
names = {0: "h", 1: "l", 2: "a", 3: "b1", 4: "b2"}
lines = ["k_accum_affine launch by launch (%d traced proofs, %d counted proofs; stand-alone, one proof at a time)" % (len(ft), len(fp)), "",
         "| launch | field | us stand-alone | VALU M wave-instr | floor us | efficiency |", "|---|---|---|---|---|---|"]
tot_d = tot_f = 0.0
for key in sorted(dur):
    if key not in ins:
        continue
    d = sum(dur[key]) / len(dur[key])
    v = sum(ins[key]) / len(ins[key])
    f = v / PEAK * 1e6
    if key[1] == "G1":
        tot_d += d; tot_f += f
    lines.append("| %s | %s | %.1f | %.1f | %.1f | %.0f%% |" % (names.get(key[0], str(key[0])), key[1], d, v / 1e6, f, 100 * f / d))
lines += ["", "G1 launches together: %.1f us stand-alone, floor %.1f us (%.0f%%)" % (tot_d, tot_f, 100 * tot_f / tot_d if tot_d else 0)]
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
