#!/usr/bin/env python3
"""Per kernel FAMILY: counters of several rocprofv3 --pmc rocpd databases (one per counter set, the same command) summed per
steady-state proof, with the fractions DESIGN.md 7.7 argues from:
  valu_busy   = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES-derived SIMD cycles) is not comparable across launches of different
                shapes, so the table gives the two ratios that are: VALU instruction cycles per WAVE cycle
                (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: how much of a resident wave's life issues VALU work) and the waits
                (SQ_WAIT_INST_ANY, SQ_WAIT_ANY) as shares of SQ_WAVE_CYCLES;
  share       = the family's share of a proof's SQ_INSTS_VALU.
usage: rocpd_families.py out.md db1 [db2 ...]"""
import re
import sqlite3
import sys

FAMILIES = [("G1 accumulation", r"k_accum_affine<|k_accum_affine_g1s"), ("G2 accumulation", r"k_accum_affine_g2"), ("transform passes", r"k_ntt29_pass"),
            ("sparse products", r"k_sell29"), ("grouping (count/place)", r"k_part_"), ("combine levels", r"k_combine_wave"),
            ("bucket reduction", r"k_bucket_|k_bit_sums"), ("witness conversion / fills / folds", r"k_w_to29|k_fill_zero|k_fold29|k_replan")]
out_path, dbs = sys.argv[1], sys.argv[2:]
fam = {}
proofs = None
for path in dbs:
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, value, start from counters_collection").fetchall()
    t_tab = max([r[3] for r in rows if "k_table_next" in r[0]] + [0])
    n_h = {}
    for n, c, v, s in rows:
        if s < t_tab:
            continue
        f = next((name for name, pat in FAMILIES if re.search(pat, n)), None)
        if f is None:
            continue
        fam.setdefault(f, {}).setdefault(c, 0.0)
        fam[f][c] += v
        if "k_w_to29" in n:
            n_h[c] = n_h.get(c, 0) + 1            # one witness conversion per proof
    if n_h:
        proofs = max(n_h.values()) if proofs is None else min(proofs, max(n_h.values()))
proofs = proofs or 1
ctrs = sorted({c for f in fam.values() for c in f})
tot_valu = sum(f.get("SQ_INSTS_VALU", 0.0) for f in fam.values()) or 1.0
hdr = ["family", "share of VALU instr"] + ["%s / proof" % c for c in ctrs] + ["VALU active / wave-cycles", "wait-inst / wave-cycles",
                                                                               "wait-any / wave-cycles", "VALU instr per wave"]
lines = ["steady-state proofs in the pass: %d (counters summed over the family's launches, divided by the proofs)" % proofs, "",
         "| " + " | ".join(hdr) + " |", "|" + "---|" * len(hdr)]
for name, _ in FAMILIES:
    f = fam.get(name)
    if not f:
        continue
    wc = f.get("SQ_WAVE_CYCLES", 0.0)
    ratio = lambda k: ("%.3f" % (f[k] / wc)) if wc and k in f else "-"
    row = [name, "%.1f %%" % (100.0 * f.get("SQ_INSTS_VALU", 0.0) / tot_valu)] + ["%.4g" % (f.get(c, 0.0) / proofs) for c in ctrs]
    row += [ratio("SQ_ACTIVE_INST_VALU"), ratio("SQ_WAIT_INST_ANY"), ratio("SQ_WAIT_ANY"),
            ("%.0f" % (f["SQ_INSTS_VALU"] / f["SQ_WAVES"])) if f.get("SQ_WAVES") and "SQ_INSTS_VALU" in f else "-"]
    lines.append("| " + " | ".join(row) + " |")
open(out_path, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
