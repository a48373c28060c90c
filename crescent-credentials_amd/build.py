"""Builds libcrescent_gpu.so (HIP, gfx950) and the bench/test workload generator in-tree.

hipcc cross-compiles gfx950 without a GPU, so this also runs in the CPU-only dev container.
Objects are cached under `_build/` and rebuilt when a source or header is newer.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libcrescent_gpu.so")
# the A/B + fault-injection build (-DCG_TUNING: the only build that reads CG_* environment switches; csrc/common.hpp).
# Same sources, objects under _build_tuning/.  tools/ab_*.sh and the fault-injection test load it through CRESCENT_GPU_LIB.
TUNING_LIB = os.path.join(HERE, "libcrescent_gpu_tuning.so")
TUNING_BUILD = os.path.join(HERE, "_build_tuning")
SYNTH_LIB = os.path.join(HERE, "libcg_synth.so")
# the reference-side caller in plain C (integration/c): built here so that every build proves the header is valid C
# and that the ABI links without Python or torch
CALLER_SRC = os.path.join(HERE, "..", "integration", "c", "crescent_prove.c")
CALLER_BIN = os.path.join(HERE, "..", "integration", "c", "crescent_prove")
# a multi-threaded C host proving in a loop (steady-state proofs/s with no Python in the process)
THROUGHPUT_SRC = os.path.join(HERE, "..", "integration", "c", "crescent_throughput.c")
THROUGHPUT_BIN = os.path.join(HERE, "..", "integration", "c", "crescent_throughput")

HIP_SOURCES = ["ntt.hip", "wmap29.hip", "msm.hip", "ecntt.hip", "prover.hip", "unit.hip", "setup.hip", "r1cs.hip", "serialize.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# A/B aid: extra compiler flags (e.g. CG_HIPCC_EXTRA="-DCG_MUL2_ONE_CHAIN"); a change of flags rebuilds every object
EXTRA_FLAGS = os.environ.get("CG_HIPCC_EXTRA", "").split()


def source_fingerprint() -> str:
    """sha256 (first 16 hex digits) over every kernel source and header plus the compiler flags in force: what a
    committed counter pass (profiles/pmc_counters.json) is valid for.  bench.py nulls the figures derived from a pass
    whose fingerprint differs from the tree it runs on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS + EXTRA_FLAGS).encode())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libcrescent_gpu cannot be built (no CPU fallback exists)")


def _newest_header() -> float:
    t = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".hpp", ".h")):
                t = max(t, os.path.getmtime(os.path.join(root, f)))
    return t


def _compile(src: str, obj: str, more=()) -> None:
    cmd = [_hipcc(), *HIPCC_FLAGS, *EXTRA_FLAGS, *more, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))


def build_tuning(verbose: bool = False, jobs: int = int(os.environ.get("CG_BUILD_JOBS", "4"))) -> str:
    """libcrescent_gpu_tuning.so: the same sources with -DCG_TUNING (environment switches compiled in)."""
    os.makedirs(TUNING_BUILD, exist_ok=True)
    hdr_t = _newest_header()
    stamp = os.path.join(TUNING_BUILD, "flags.txt")
    flags_now = " ".join(HIPCC_FLAGS + EXTRA_FLAGS + ["-DCG_TUNING"])
    flags_changed = (open(stamp).read() if os.path.exists(stamp) else "") != flags_now
    todo, objs = [], []
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(TUNING_BUILD, s.replace(".hip", ".o"))
        objs.append(obj)
        if flags_changed or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append((src, obj))
    if todo:
        if verbose:
            print("[build:tuning] compiling", [os.path.basename(s) for s, _ in todo], file=sys.stderr)
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            for f in [ex.submit(_compile, s, o, ("-DCG_TUNING",)) for s, o in todo]:
                f.result()
        with open(stamp, "w") as f:
            f.write(flags_now)
    if todo or not os.path.exists(TUNING_LIB):
        r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TUNING_LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return TUNING_LIB


def build(verbose: bool = False, jobs: int = int(os.environ.get("CG_BUILD_JOBS", "4"))) -> str:
    os.makedirs(BUILD, exist_ok=True)
    hdr_t = _newest_header()
    stamp = os.path.join(BUILD, "flags.txt")
    flags_now = " ".join(HIPCC_FLAGS + EXTRA_FLAGS)
    flags_changed = (open(stamp).read() if os.path.exists(stamp) else " ".join(HIPCC_FLAGS)) != flags_now
    todo = []
    objs = []
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(BUILD, s.replace(".hip", ".o"))
        objs.append(obj)
        if flags_changed or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append((src, obj))
    if todo:
        if verbose:
            print("[build] compiling", [os.path.basename(s) for s, _ in todo], file=sys.stderr)
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            for f in [ex.submit(_compile, s, o) for s, o in todo]:
                f.result()
        with open(stamp, "w") as f:
            f.write(flags_now)
    if todo or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    # workload generator: host-only C++ (no GPU code), used by tests/bench to make synthetic circuits
    synth_src = os.path.join(HERE, "synth", "synth.cpp")
    if not os.path.exists(SYNTH_LIB) or os.path.getmtime(SYNTH_LIB) < max(os.path.getmtime(synth_src), hdr_t):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", SYNTH_LIB, synth_src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("synth build failed:\n" + r.stderr[-4000:])
    if os.path.exists(CALLER_SRC) and (todo or not os.path.exists(CALLER_BIN) or
                                       os.path.getmtime(CALLER_BIN) < max(os.path.getmtime(CALLER_SRC), hdr_t)):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "lib")
        cmd = ["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-pthread",
               "-I", os.path.join(HERE, "..", "include"), CALLER_SRC, "-o", CALLER_BIN, "-L", HERE, "-lcrescent_gpu",
               "-Wl,-rpath,$ORIGIN/../../crescent-credentials_amd", "-Wl,-rpath-link," + rocm_lib, "-Wl,-rpath," + rocm_lib]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("integration/c/crescent_prove build failed:\n" + r.stderr[-4000:])
    if os.path.exists(THROUGHPUT_SRC) and (todo or not os.path.exists(THROUGHPUT_BIN) or
                                           os.path.getmtime(THROUGHPUT_BIN) < max(os.path.getmtime(THROUGHPUT_SRC), hdr_t,
                                                                                  os.path.getmtime(SYNTH_LIB))):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "lib")
        cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-pthread",
               "-I", os.path.join(HERE, "..", "include"), THROUGHPUT_SRC, "-o", THROUGHPUT_BIN, "-L", HERE, "-lcrescent_gpu", "-lcg_synth",
               "-Wl,-rpath,$ORIGIN/../../crescent-credentials_amd", "-Wl,-rpath-link," + rocm_lib, "-Wl,-rpath," + rocm_lib]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("integration/c/crescent_throughput build failed:\n" + r.stderr[-4000:])
    return LIB


UBENCH_SRC = os.path.join(HERE, "..", "tools", "ubench", "valu_rates.hip")
UBENCH_BIN = os.path.join(HERE, "..", "tools", "ubench", "valu_rates")


def build_ubench(verbose: bool = False) -> str:
    """tools/ubench/valu_rates: the VALU issue-rate micro-benchmark bench.py runs (--json) to price `roofline_valu` in SIMD
    cycles measured on the box it runs on"""
    if not os.path.exists(UBENCH_BIN) or os.path.getmtime(UBENCH_BIN) < max(os.path.getmtime(UBENCH_SRC), _newest_header()):
        if verbose:
            print("[build] compiling tools/ubench/valu_rates", file=sys.stderr)
        r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O2", "-o", UBENCH_BIN, UBENCH_SRC], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("valu_rates build failed:\n" + r.stderr[-4000:])
    return UBENCH_BIN


def write_isa_class_counts(verbose: bool = False) -> str:
    """profiles/isa_class_counts.json: per-kernel static VALU instruction-class counts from the code objects of the library
    just built (tools/isa_mix.py), stamped with the source fingerprint; rewritten only when the fingerprint changed"""
    import json
    out = os.path.join(HERE, "..", "profiles", "isa_class_counts.json")
    try:
        with open(out) as f:
            if json.load(f).get("csrc_sha16") == source_fingerprint():
                return out
    except Exception:
        pass
    r = subprocess.run([sys.executable, os.path.join(HERE, "..", "tools", "isa_mix.py"), LIB, out], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("tools/isa_mix.py failed:\n" + (r.stdout + r.stderr)[-3000:])
    if verbose:
        print("[build]", r.stdout.strip().splitlines()[0], file=sys.stderr)
    return out


if __name__ == "__main__":
    print(build(verbose=True))
    if "--tuning" in sys.argv[1:]:
        print(build_tuning(verbose=True))
