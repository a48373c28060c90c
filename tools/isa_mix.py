#!/usr/bin/env python3
"""Static VALU instruction-class counts of every kernel in libcrescent_gpu.so, from its own gfx950 code objects.

bench.py prices the prove path's vector-ALU work in SIMD cycles: (wave-instructions of a class) x (the cycles the SIMD
takes to issue one, measured in-run by tools/ubench/valu_rates --json).  The dynamic instruction COUNT per kernel comes
from the committed SQ_INSTS_VALU pass (profiles/pmc_counters.json); how a kernel's instructions split over the classes comes
from here: the disassembly of the library the bench loads, not a hand count.

Classes (as the micro-benchmark measures them):
  mad64    v_mad_u64_u32 / v_mad_i64_i32 - the 32 x 32 + 64 multiply-adds the limb products are made of
  simple32 VALU instructions in a 32-bit encoding (VOP1 / VOP2: the disassembler's _e32 suffix) - mov, and, or, add, sub,
           shifts, cndmask
  other    every other VALU instruction (VOP3 / _e64 encodings, 64-bit shifts and adds, carry chains, v_readlane ...)
Only the kernel's hot part would be the right sample of a kernel whose prologue is large; the kernels that carry the
instruction count are loop bodies unrolled hundreds of times (a mixed addition is ~2200 instructions), so the whole body
is the sample.

usage: tools/isa_mix.py [path/to/libcrescent_gpu.so] [out.json]     (default out: profiles/isa_class_counts.json)
"""
import collections
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAD64 = ("v_mad_u64_u32", "v_mad_i64_i32")


def classify(mnemonic: str):
    if not mnemonic.startswith("v_"):
        return None
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", mnemonic)
    if base in MAD64:
        return "mad64"
    if mnemonic.endswith("_e32"):
        return "simple32"
    return "other"


def demangle(names):
    """-> the names as rocprofv3 prints them: demangled, without the return type and the argument list"""
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        except Exception:
            continue
        short = {}
        for n, d in zip(names, out):
            d = re.sub(r"^void ", "", d)
            depth, cut = 0, len(d)
            for i, ch in enumerate(d):              # the argument list opens at the first '(' outside template brackets
                if ch == "<":
                    depth += 1
                elif ch == ">":
                    depth -= 1
                elif ch == "(" and depth == 0:
                    cut = i
                    break
            short[n] = d[:cut]
        return short
    return {n: n for n in names}


def kernel_counts(lib_path):
    tmp = tempfile.mkdtemp(prefix="isa_mix_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, so)
        subprocess.run([OBJDUMP, "--offloading", so], cwd=tmp, capture_output=True, check=True)
        counts = {}
        for co in sorted(glob.glob(so + ".*gfx950")):
            if os.path.getsize(co) == 0:
                continue
            # kernels are the symbols that have a .kd descriptor
            syms = subprocess.run([OBJDUMP, "-t", co], capture_output=True, text=True).stdout
            kernels = {m.group(1) for m in re.finditer(r"\s(\S+)\.kd\s*$", syms, flags=re.M)}
            dis = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True).stdout
            cur = None
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1) if m.group(1) in kernels else None
                    if cur:
                        counts[cur] = collections.Counter()
                    continue
                if cur is None:
                    continue
                m = re.match(r"^\s+(\S+)", line)
                if not m:
                    continue
                c = classify(m.group(1))
                if c:
                    counts[cur][c] += 1
                counts[cur]["all_instructions"] += 1
        return counts
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "crescent-credentials_amd", "libcrescent_gpu.so")
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "isa_class_counts.json")
    sys.path.insert(0, os.path.join(ROOT, "crescent-credentials_amd"))
    import build as cg_build
    counts = kernel_counts(lib)
    names = demangle(sorted(counts))
    kernels = {}
    for k in sorted(counts):
        c = counts[k]
        valu = c["mad64"] + c["simple32"] + c["other"]
        if not valu:
            continue
        kernels[names.get(k, k)] = {"valu": valu, "mad64": c["mad64"], "simple32": c["simple32"], "other": c["other"],
                                   "all_instructions": c["all_instructions"]}
    rec = {"csrc_sha16": cg_build.source_fingerprint(), "library": os.path.relpath(lib, ROOT),
           "classes": {"mad64": list(MAD64), "simple32": "VALU in a 32-bit encoding (_e32)", "other": "every other VALU instruction"},
           "kernels": kernels}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("wrote %s: %d kernels" % (out, len(kernels)))
    for name in ("k_accum_affine_g1s", "k_accum_affine_g2", "k_ntt29_pass", "k_bucket_chunks", "k_sell29"):
        for k, v in kernels.items():
            if name in k:
                print("  %-70s valu %6d  mad64 %.2f  simple32 %.2f  other %.2f" % (k[:70], v["valu"], v["mad64"] / v["valu"], v["simple32"] / v["valu"],
                                                                                    v["other"] / v["valu"]))


if __name__ == "__main__":
    main()
