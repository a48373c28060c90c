#!/usr/bin/env python3
"""Where the issue slots of the PIPELINED run go: a rocprofv3 --kernel-trace database of the default bench (16 proofs in
flight) reduced to

  1. per 100 us of the steady window: how much of the chip's resident-wave capacity the kernels in flight ASK for
     (sum over kernels running in the bin of min(their waves, what fits of them) x their VGPR allocation, over
     256 CUs x 4 SIMDs x 512 VGPRs), as a histogram - a chip that is asked for less than it holds is idle for lack of
     work (launch gaps, tails), a chip that is asked for more is as full as the kernels' own stalls allow;
  2. which kernels hold that capacity, time-weighted;
  3. per kernel family: launches per proof, mean duration in the pipeline, and the time it spends ALONE on the chip
     or with less than a quarter of the capacity asked for.

usage: rocpd_pipeline.py results.db [proofs_in_window]   (writes markdown to stdout)
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]


def col(*names):
    for n in names:
        if n in cols:
            return n
    return None


c_grid = col("grid_x", "grid_size_x", "grid_size")
c_wg = col("workgroup_x", "workgroup_size_x", "workgroup_size")
c_gy, c_gz = col("grid_y", "grid_size_y"), col("grid_z", "grid_size_z")
c_vgpr = col("arch_vgpr_count", "vgpr_count")
c_lds = col("lds_block_size", "lds_size", "group_segment_size")
sel = ["name", "start", "end"] + [c or "0" for c in (c_grid, c_gy, c_gz, c_wg, c_vgpr, c_lds)]
rows = db.execute("select %s from kernels order by start" % ", ".join(sel)).fetchall()
if not rows:
    sys.exit("no kernels in the trace")
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("cg::", "")[:60]
# steady window: from the last window-table build (the re-tune) + 15 % of what follows, to 95 %
t_tab = max([r[2] for r in rows if "k_table_next" in r[0]] + [rows[0][1]])
t_end = max(r[2] for r in rows)
w0, w1 = t_tab + 0.15 * (t_end - t_tab), t_tab + 0.95 * (t_end - t_tab)
K = []
for name, s, e, gx, gy, gz, wg, vgpr, lds in rows:
    if e <= w0 or s >= w1:
        continue
    threads = max(1, int(gx or 1)) * max(1, int(gy or 1)) * max(1, int(gz or 1))
    wg = max(1, int(wg or 64))
    waves = (threads + 63) // 64
    vg = int(vgpr or 64)
    vg_alloc = ((vg + 7) // 8) * 8
    per_simd_by_vgpr = max(1, min(8, 512 // max(8, vg_alloc)))
    waves_per_block = (wg + 63) // 64
    lds = int(lds or 0)
    blocks_per_cu = 32 if lds == 0 else max(1, (160 * 1024) // max(1, lds))
    fit = min(per_simd_by_vgpr * 4 * 256, blocks_per_cu * waves_per_block * 256)     # waves of this kernel the chip can hold
    K.append((short(name), max(s, w0), min(e, w1), min(waves, fit) * vg_alloc, e - s))
CAP = 256 * 4 * 512.0
BIN = 100_000   # ns
nb = int((w1 - w0) // BIN) + 1
ask = [0.0] * nb
who = {}
for name, s, e, demand, _ in K:
    b0, b1 = int((s - w0) // BIN), int((e - w0 - 1) // BIN)
    for b in range(b0, b1 + 1):
        lo, hi = w0 + b * BIN, w0 + (b + 1) * BIN
        f = (min(e, hi) - max(s, lo)) / BIN
        ask[b] += f * demand / CAP
        who[name] = who.get(name, 0.0) + f * min(demand / CAP, 1.0)
ask = ask[:-1] or ask
proofs = float(sys.argv[2]) if len(sys.argv) > 2 else None
if proofs is None:   # the witness map converts the assignment once per proof
    proofs = float(sum(1 for k in K if k[0].startswith("k_w_to29")))
print("# pipelined run: resident-wave capacity asked for, per 100 us of the steady window\n")
print("window %.1f ms, %d kernels, ~%.0f proofs (%.1f launches per proof)\n" % ((w1 - w0) / 1e6, len(K), proofs, len(K) / max(1.0, proofs)))
edges = [0.0, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0, 1e9]
hist = [0] * (len(edges) - 1)
for a in ask:
    for i in range(len(edges) - 1):
        if edges[i] <= a < edges[i + 1]:
            hist[i] += 1
print("| capacity asked for (x the chip's VGPR file) | share of the window |\n|---|---|")
for i, h in enumerate(hist):
    hi = "%.2f" % edges[i + 1] if edges[i + 1] < 1e8 else "more"
    print("| %.2f - %s | %.1f %% |" % (edges[i], hi, 100.0 * h / len(ask)))
print("\nmean asked %.2f, median %.2f; bins below 1.0: %.1f %% of the window\n" % (sum(ask) / len(ask), sorted(ask)[len(ask) // 2],
                                                                                 100.0 * sum(1 for a in ask if a < 1.0) / len(ask)))
print("| kernel | time-weighted share of the capacity held (each capped at the whole chip) |\n|---|---|")
tot = sum(who.values())
for n, v in sorted(who.items(), key=lambda kv: -kv[1])[:16]:
    print("| %s | %.1f %% |" % (n, 100.0 * v / tot))
# per family: launches per proof, mean duration, time spent with the chip mostly empty around it
ev = []
for i, (name, s, e, demand, _) in enumerate(K):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active, last, lonely = {}, ev[0][0], {}
for t, kind, i in ev:
    if active and t > last:
        total = sum(active.values()) / CAP
        if total < 0.25:
            for j in active:
                lonely[K[j][0]] = lonely.get(K[j][0], 0.0) + (t - last)
    last = t
    if kind == 1:
        active[i] = K[i][3]
    else:
        active.pop(i, None)
fam = {}
for name, s, e, demand, dur in K:
    a = fam.setdefault(name, [0, 0.0])
    a[0] += 1; a[1] += dur
print("\n| kernel | launches per proof | mean us in the pipeline | ms per proof | of which with < 1/4 of the chip asked for |\n|---|---|---|---|---|")
for n, (cnt, dur) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]:
    print("| %s | %.2f | %.1f | %.3f | %.3f |" % (n, cnt / proofs, dur / cnt / 1e3, dur / proofs / 1e6, lonely.get(n, 0.0) / proofs / 1e6))
print("\nsum of kernel durations per proof: %.2f ms; window per proof: %.3f ms (the ratio is the mean number of kernels in flight)" %
      (sum(v[1] for v in fam.values()) / proofs / 1e6, (w1 - w0) / proofs / 1e6))
