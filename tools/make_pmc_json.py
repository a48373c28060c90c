#!/usr/bin/env python3
"""profiles/pmc_counters.json entry for one workload from three rocprofv3 --pmc passes of the same bench.py command
(FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU; one counter per pass as guides/MI355X_MICROARCH.md prescribes).
usage: make_pmc_json.py <workload-key> fetch.db write.db valu.db <command string> [out.json]"""
import json, os, re, sqlite3, sys


def rows(db, ctr):
    r = sqlite3.connect(db).execute("select kernel_name, counter_name, value, start, dispatch_id from counters_collection").fetchall()
    t_tab = max([x[3] for x in r if "k_table_next" in x[0]] + [0])
    first = min(x[3] for x in r if "k_w_to29" in x[0] and x[3] > t_tab)
    r = [x for x in r if x[3] >= first and x[1] == ctr]
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
}
print(json.dumps({key: entry}, indent=1))
if out_path:
    cur = {}
    if os.path.exists(out_path):
        cur = json.load(open(out_path))
    cur[key] = entry
    json.dump(cur, open(out_path, "w"), indent=1)
