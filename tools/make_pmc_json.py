#!/usr/bin/env python3
"""profiles/pmc_counters.json entry for one workload from three rocprofv3 --pmc passes of the same bench.py command
(FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU; one counter per pass as guides/MI355X_MICROARCH.md prescribes).
usage: make_pmc_json.py <workload-key> fetch.db write.db valu.db <command string> [out.json]"""
import json, os, re, sqlite3, sys


def short(name):
    """a kernel's name as tools/isa_mix.py writes it: no return type, no argument list"""
    d = re.sub(r"^void ", "", name)
    depth = 0
    for i, ch in enumerate(d):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return d[:i]
    return d


def rows(db, ctr):
    r = sqlite3.connect(db).execute("select kernel_name, counter_name, value, start, dispatch_id from counters_collection").fetchall()
    t_tab = max([x[3] for x in r if "k_table_next" in x[0]] + [0])
    first = min(x[3] for x in r if "k_w_to29" in x[0] and x[3] > t_tab)
    r = [(short(x[0]),) + tuple(x[1:]) for x in r if x[3] >= first and x[1] == ctr]
    nproofs = len(set(x[4] for x in r if "k_w_to29" in x[0]))
    return r, nproofs


key, fdb, wdb, vdb, cmd = sys.argv[1:6]
out_path = sys.argv[6] if len(sys.argv) > 6 else None
is_acc = lambda n: ("k_accum_affine<" in n and "Fq2_29" not in n) or "k_accum_affine_g1s" in n     # the G1 accumulation, either form
f, nf = rows(fdb, "FETCH_SIZE")
w, nw = rows(wdb, "WRITE_SIZE")
v, nv = rows(vdb, "SQ_INSTS_VALU")
fa = [x[2] for x in f if is_acc(x[0])]
wa = [x[2] for x in w if is_acc(x[0])]
fetch_kb = sum(fa) / len(fa)
write_kb = sum(wa) / len(wa)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "crescent-credentials_amd"))
import build as cg_build   # noqa: E402  (the package directory has a hyphen: imported by path)
# ---- per kernel, per steady-state proof: what bench.py weights the static instruction classes with (profiles/isa_class_counts.json)
def per_kernel(rws, n):
    d = {}
    for x in rws:
        d[x[0]] = d.get(x[0], 0.0) + x[2]
    return {k: round(v / n, 1) for k, v in sorted(d.items(), key=lambda kv: -kv[1])}


valu_k, fetch_k, write_k = per_kernel(v, nv), per_kernel(f, nf), per_kernel(w, nw)
# ---- FETCH_SIZE / WRITE_SIZE calibrated on a streaming kernel of known bytes (guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE is
# half the bytes of a wide coalesced read on gfx950, WRITE_SIZE and other widths uncalibrated).  The first pass of a transform,
# k_ntt29_pass<0, 0, B>, reads its D x 32-byte vector once and writes it once (its twiddles are the 2^(B-1) of a tile: on-chip),
# D = the domain the bench proves on.
cal = None
ntt_f = [x[2] for x in f if re.match(r"cg::k_ntt29_pass<0, 0, \d+>", x[0])]
ntt_w = [x[2] for x in w if re.match(r"cg::k_ntt29_pass<0, 0, \d+>", x[0])]
m_dom = re.search(r"--shape (\S+)", cmd)
log_d = 22 if (m_dom and m_dom.group(1) in ("mdl1", "rs256-sd-large")) else 21
if ntt_f and ntt_w:
    known_kb = (1 << log_d) * 32 / 1024.0
    cal = {"kernel": "cg::k_ntt29_pass<0, 0, *>", "known_kb_read_and_written_per_launch": known_kb,
           "fetch_kb_counted": round(sum(ntt_f) / len(ntt_f), 1), "write_kb_counted": round(sum(ntt_w) / len(ntt_w), 1),
           "read_factor": round(known_kb / (sum(ntt_f) / len(ntt_f)), 4), "write_factor": round(known_kb / (sum(ntt_w) / len(ntt_w)), 4)}
# access patterns: the factor applies to kernels that stream their operands in wide coalesced accesses; the accumulations gather
# 64-byte table points (factor 1: profiles/r01_r_fetch_size_calibration.txt); everything else is left raw and listed as such
STREAM = ("k_ntt29_pass", "k_fill_zero", "k_w_to29", "k_table_next", "k_ec_stage", "k_fold29", "k_pointwise", "k_shard_major")
GATHER64 = ("k_accum_affine",)


def corrected(per_k, factor):
    tot, by = 0.0, {"stream": 0.0, "gather64": 0.0, "raw": 0.0}
    for k, kb in per_k.items():
        if any(s in k for s in STREAM):
            tot += kb * factor; by["stream"] += kb * factor
        elif any(s in k for s in GATHER64):
            tot += kb; by["gather64"] += kb
        else:
            tot += kb; by["raw"] += kb
    return round(tot, 1), {k: round(x, 1) for k, x in by.items()}


fetch_corr = write_corr = None
if cal:
    fetch_corr = corrected(fetch_k, cal["read_factor"])
    write_corr = corrected(write_k, cal["write_factor"])
entry = {
    "command": cmd,
    # the kernel sources + compiler flags these counts were taken on; bench.py nulls what it derives from them when the
    # tree it runs on differs
    "csrc_sha16": cg_build.source_fingerprint(),
    "steady_state_proofs": {"fetch": nf, "write": nw, "valu": nv},
    "accum_affine_g1_launches_measured": len(fa),
    "accum_affine_g1_fetch_kb_per_launch": round(fetch_kb, 1),
    "accum_affine_g1_write_kb_per_launch": round(write_kb, 1),
    # the raw FETCH_SIZE is the byte count for this kernel's access pattern (random 64-byte gathers + an 8-byte record
    # stream): calibrated on a known count of such gathers, profiles/r01_r_fetch_size_calibration.txt; the x2 of the
    # guide applies to wide coalesced streams only
    "accum_affine_g1_hbm_bytes_per_launch": round((fetch_kb + write_kb) * 1024.0),
    "valu_wave_instr_per_proof": round(sum(x[2] for x in v) / nv),
    "fetch_kb_per_proof_all_kernels": round(sum(x[2] for x in f) / nf, 1),
    "write_kb_per_proof_all_kernels": round(sum(x[2] for x in w) / nw, 1),
    "valu_wave_instr_per_proof_by_kernel": valu_k,
    "stream_calibration": cal,
    "fetch_kb_per_proof_corrected": fetch_corr[0] if fetch_corr else None,
    "fetch_kb_per_proof_corrected_by_pattern": fetch_corr[1] if fetch_corr else None,
    "write_kb_per_proof_corrected": write_corr[0] if write_corr else None,
    "write_kb_per_proof_corrected_by_pattern": write_corr[1] if write_corr else None,
    "fetch_kb_per_proof_upper_bound": round(sum(kb * (1.0 if any(g in k for g in GATHER64) else 2.0) for k, kb in fetch_k.items()), 1),
    "ntt_first_pass_hbm_kb_per_launch_corrected": round(cal["fetch_kb_counted"] * cal["read_factor"] + cal["write_kb_counted"] * cal["write_factor"], 1) if cal else None,
    "correction_note": "corrected = counter x the factor measured on k_ntt29_pass<0,0,*> (known bytes) for the kernels that stream wide "
                       "coalesced operands, x 1 for the accumulations' 64-byte gathers (r01 calibration), raw for the rest; upper bound = "
                       "every non-gather kernel's fetch doubled (the guide's factor for wide reads)",
}
print(json.dumps({key: entry}, indent=1))
if out_path:
    cur = {}
    if os.path.exists(out_path):
        cur = json.load(open(out_path))
    cur[key] = entry
    json.dump(cur, open(out_path, "w"), indent=1)
