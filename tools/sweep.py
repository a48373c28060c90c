#!/usr/bin/env python3
"""Unit sweeps of SURVEY.md §8(d): MSM over n = 2^k resident bases (G1, G2) and the NTT over 2^k elements, through the
handle entry points of the C ABI (cg_msm_load_*/cg_msm_run, cg_ntt_load/cg_ntt_run), operands resident in HBM.

Reports, per size: milliseconds per call (host wall clock around the synchronous call, best of `--reps`), the
HIP-event time of the kernels, pairs/s (= G1/G2 MSM scalar-adds/s as BASELINE.json defines them) or elements/s, and
the algorithmic HBM rate of §8(d) (96 B per G1 pair, 160 B per G2 pair, 64 B per element per transform) over 8 TB/s.
Scalars are uniform below 0x30·2^248 (< r); `--bits` gives the 0/1-heavy population of circom witnesses instead.

usage: python tools/sweep.py [--g1 10:22] [--g2 10:20] [--ntt 10:24] [--reps 5] [--out file.md]"""
import argparse, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import crescent_credentials_amd as cc

HBM_PEAK = 8e12


def rng_range(s):
    a, b = s.split(":")
    return range(int(a), int(b) + 1)


def random_scalars(n, gen, bit_fraction):
    """n x 32 B canonical scalars on the device"""
    t = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen)
    t[:, 31] = t[:, 31] % 0x30
    if bit_fraction > 0:
        u = torch.rand(n, device="cuda", generator=gen)
        zero = u < bit_fraction / 2
        one = (u >= bit_fraction / 2) & (u < bit_fraction)
        t[zero | one] = 0
        t[one, 0] = 1
    return t.contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--g1", default="10:22")
    ap.add_argument("--g2", default="10:20")
    ap.add_argument("--ntt", default="10:24")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--bits", type=float, default=0.0, help="fraction of scalars that are 0 or 1 (0.9 = circom-like)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    assert torch.cuda.is_available(), "needs a GPU"
    assert cc.lib().cg_init(0, None) == 0
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0xC5E5CE47)
    lines = ["# unit sweeps (%s scalars; operands resident in HBM; best of %d calls)" % ("uniform" if a.bits == 0 else "%.0f%% 0/1" % (100 * a.bits), a.reps), ""]

    for group, spec, pair_bytes in ((1, a.g1, 96), (2, a.g2, 160)):
        ks = list(rng_range(spec)) if spec else []
        if not ks:
            continue
        nmax = 1 << max(ks)
        t0 = time.time()
        seeds = np.random.default_rng(group).integers(0, 256, (nmax, 32), dtype=np.uint8)
        seeds[:, 31] %= 0x30                                   # < r
        seeds[:, 0] |= 1                                       # non-zero: no identity bases
        bases = (cc.fixed_base_g1 if group == 1 else cc.fixed_base_g2)(seeds)
        print("[sweep] G%d: %d bases made in %.1fs" % (group, nmax, time.time() - t0), file=sys.stderr, flush=True)
        lines += ["## MSM G%d" % group, "",
                  "| n | entries (mixed additions) | ms/call (wall) | ms kernels | ms accumulate | pairs/s | algorithmic GB/s | of 8 TB/s |", "|---|---|---|---|---|---|---|---|"]
        pb = 64 * group
        for k in ks:
            n = 1 << k
            ctx = cc.MsmContext(bases[:n * pb], group=group)
            sc = random_scalars(n, gen, a.bits)
            torch.cuda.synchronize()
            best = None
            for rep in range(a.reps + 1):
                t = time.perf_counter()
                _, tm = ctx.run_dev(sc.data_ptr(), n, timings=True)
                dt = time.perf_counter() - t
                if rep and (best is None or dt < best[0]):
                    best = (dt, tm)
            dt, tm = best
            kern = tm["msm_h_ms"] if group == 1 else tm["msm_b2_ms"]
            acc = tm["accum_g1_ms"] if group == 1 else tm["accum_g2_ms"]
            ent = tm["entries_g1"] if group == 1 else tm["entries_g2"]
            rate = n / dt
            lines.append("| 2^%d | %d | %.3f | %.3f | %.3f | %.3e | %.1f | %.2f%% |" %
                         (k, ent, dt * 1e3, kern, acc, rate, rate * pair_bytes / 1e9, 100 * rate * pair_bytes / HBM_PEAK))
            ctx.close()
            del sc
        lines.append("")

    ks = list(rng_range(a.ntt)) if a.ntt else []
    if ks:
        lines += ["## NTT (forward, coset; Fr)", "", "| n | ms/call (wall) | ms kernels | elements/s | algorithmic GB/s | of 8 TB/s |", "|---|---|---|---|---|---|"]
        for k in ks:
            n = 1 << k
            ctx = cc.NttContext(k)
            d = random_scalars(n, gen, 0.0)
            torch.cuda.synchronize()
            best = None
            for rep in range(a.reps + 1):
                t = time.perf_counter()
                ms = ctx.run_dev(d.data_ptr(), inverse=bool(rep & 1), coset=True)
                dt = time.perf_counter() - t
                if rep and (best is None or ms < best[1]):
                    best = (dt, ms)
            dt, ms = best
            rate = n / (ms * 1e-3)
            lines.append("| 2^%d | %.3f | %.3f | %.3e | %.1f | %.2f%% |" % (k, dt * 1e3, ms, rate, rate * 64 / 1e9, 100 * rate * 64 / HBM_PEAK))
            ctx.close()
            del d
        lines.append("")
    out = "\n".join(lines)
    print(out)
    if a.out:
        with open(a.out, "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
