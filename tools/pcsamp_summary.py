#!/usr/bin/env python3
"""Reduces rocprofv3 stochastic PC-sampling output (csv) of a pipelined bench.py run: per kernel family, how many samples,
what share of them ISSUED an instruction (and of which type), and why the others stalled.  Written without having seen
the beta's output: it groups by every categorical column it recognises and reports the columns it found.
usage: pcsamp_summary.py <raw-dir> <out.md>"""
import collections
import csv
import glob
import os
import re
import sys

raw, out_path = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
files = glob.glob(os.path.join(raw, "**", "*.csv"), recursive=True)
ktrace = [f for f in files if "kernel_trace" in f]
samples = [f for f in files if "pc_sampling" in f]
lines = ["files: " + ", ".join(os.path.basename(f) for f in files)]
name_of = {}
for f in ktrace:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            did = r.get("Dispatch_Id") or r.get("dispatch_id")
            nm = r.get("Kernel_Name") or r.get("kernel_name") or ""
            nm = re.sub(r"\(.*", "", nm).replace("void ", "").replace("cg::", "")
            nm = re.sub(r"F29<Fq29P>\s*", "Fq29", nm)
            if did:
                name_of[did] = nm[:60]
CAT = ("wave_issued", "Wave_Issued", "instruction_type", "Instruction_Type", "stall_reason", "Stall_Reason", "snapshot_stall_reason",
       "wave_count", "Wave_Count")
for f in samples:
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames or []
        lines.append("")
        lines.append("## %s" % os.path.basename(f))
        lines.append("columns: " + ", ".join(cols))
        cats = [c for c in cols if c in CAT or c.lower().startswith(("arb_state", "snapshot_", "stall", "inst_type", "wave_issued"))]
        dcol = next((c for c in cols if c.lower() in ("dispatch_id", "correlation_id_internal")), None)
        per = collections.defaultdict(lambda: collections.defaultdict(collections.Counter))
        tot = collections.Counter()
        for r in rd:
            k = name_of.get(r.get(dcol, ""), "?") if dcol else "?"
            tot[k] += 1
            for c in cats:
                per[k][c][r.get(c, "")] += 1
        n_all = sum(tot.values())
        lines.append("samples: %d" % n_all)
        allk = collections.defaultdict(collections.Counter)
        for k in per:
            for c in per[k]:
                allk[c].update(per[k][c])
        lines.append("")
        lines.append("### all kernels")
        for c in cats:
            lines.append("* %s: " % c + ", ".join("%s %.1f %%" % (v or "-", 100.0 * n / max(1, n_all)) for v, n in allk[c].most_common(12)))
        lines.append("")
        lines.append("### by kernel (share of all samples; then per column the distribution inside the kernel)")
        for k, n in tot.most_common(40):
            lines.append("* **%s** %.1f %% of samples (%d)" % (k, 100.0 * n / max(1, n_all), n))
            for c in cats:
                lines.append("    * %s: " % c + ", ".join("%s %.1f %%" % (v or "-", 100.0 * m / n) for v, m in per[k][c].most_common(8)))
open(out_path, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
