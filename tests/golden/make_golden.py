#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory.

Two kinds of data end up here:
  * `reference_kats.json` — known-answer DATA held by the reference's own tests (byte vectors,
    decimal coordinates, a hex .r1cs file and the fields its test asserts), transcribed as values:
      forks/circom-compat/src/zkey.rs:397-460          (Fq Montgomery one, G1/G2 generator bytes)
      forks/circom-compat/src/circom/r1cs_reader.rs:183 (Fr modulus LE bytes)
      forks/circom-compat/src/circom/r1cs_reader.rs:264-345 (sample .r1cs + asserted fields)
      forks/circom-compat/src/witness/witness_calculator.rs:464-467 (Fr modulus hex)
  * `groth16_*.json`, `ntt.json`, `msm.json` — inputs + expected outputs computed by the pure-Python
    oracle (oracle/bn254_oracle.py) from seeds.  The reference itself (Rust, un-vendored arkworks)
    cannot run in this environment and pins no proof bytes, so these pin the ORACLE (and through it
    the HIP path), not arkworks: "parity unpinned" (see DESIGN.md §3).

Run from the repo root:  python tests/golden/make_golden.py
The synthetic `tiny` circuit needs the workload generator (python -c "import __graft_entry__ as g; g.build()").
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)

import bn254_oracle as o  # noqa: E402


def hx(b: bytes) -> str:
    return b.hex()


def pk_to_json(pk):
    return dict(
        alpha_g1=hx(o.g1_packed(pk["vk"]["alpha_g1"])), beta_g1=hx(o.g1_packed(pk["beta_g1"])),
        delta_g1=hx(o.g1_packed(pk["delta_g1"])), beta_g2=hx(o.g2_packed(pk["vk"]["beta_g2"])),
        gamma_g2=hx(o.g2_packed(pk["vk"]["gamma_g2"])), delta_g2=hx(o.g2_packed(pk["vk"]["delta_g2"])),
        gamma_abc_g1=hx(b"".join(o.g1_packed(p) for p in pk["vk"]["gamma_abc_g1"])),
        a_query=hx(b"".join(o.g1_packed(p) for p in pk["a_query"])),
        b_g1_query=hx(b"".join(o.g1_packed(p) for p in pk["b_g1_query"])),
        b_g2_query=hx(b"".join(o.g2_packed(p) for p in pk["b_g2_query"])),
        h_query=hx(b"".join(o.g1_packed(p) for p in pk["h_query"])),
        l_query=hx(b"".join(o.g1_packed(p) for p in pk["l_query"])),
    )


def pk_digest(pk):
    j = pk_to_json(pk)
    return {k: o.sha256_hex(bytes.fromhex(v)) for k, v in j.items()}


def groth16_case(name, matrices, l, m, M, w, seed, store_pk, rs_list):
    rng = random.Random(seed)
    trap = [rng.randrange(1, o.R) for _ in range(4)]  # tau, alpha, beta, delta
    pk, qap = o.generate_parameters(matrices, l, m, M, *trap)
    h = o.witness_map_from_matrices(matrices, l, m, w)
    proofs = []
    for r, s in rs_list:
        pr = o.create_proof_with_reduction_and_matrices(pk, r, s, matrices, l, m, w)
        assert pr == o.closed_form_proof(qap, trap, r % o.R, s % o.R, h, w, l), "closed form mismatch"
        assert o.verify_proof(pk["vk"], pr, w[1:l]), "oracle proof does not verify"
        proofs.append(dict(r=hex(r), s=hex(s), proof=hx(o.proof_uncompressed(pr)), proof_compressed=hx(o.proof_compressed(pr))))
    out = dict(
        name=name, num_inputs=l, num_constraints=m, num_variables=M, domain_size=qap["D"],
        trapdoor=dict(tau=hex(trap[0]), alpha=hex(trap[1]), beta=hex(trap[2]), delta=hex(trap[3])),
        witness=[hex(x) for x in w],
        h_sha256=o.sha256_hex(b"".join(o.fe_bytes(x) for x in h)),
        pk_sha256=pk_digest(pk),
        proofs=proofs,
    )
    if store_pk:
        out["pk"] = pk_to_json(pk)
        out["h"] = [hex(x) for x in h]
    return out


def rows_json(matrices):
    return [[[[hex(c), col] for c, col in row] for row in mat] for mat in matrices]


def main():
    rng = random.Random(20250620)

    # ---- reference KATs (data transcribed from the reference's tests) --------------------------------
    kats = dict(
        fq_montgomery_one_le=[157, 13, 143, 197, 141, 67, 93, 211, 61, 11, 199, 245, 40, 235, 120, 10, 44, 70, 121, 120,
                              111, 163, 110, 102, 47, 223, 7, 154, 193, 119, 10, 14],
        g1_generator_montgomery_le=[157, 13, 143, 197, 141, 67, 93, 211, 61, 11, 199, 245, 40, 235, 120, 10, 44, 70, 121,
                                    120, 111, 163, 110, 102, 47, 223, 7, 154, 193, 119, 10, 14, 58, 27, 30, 139, 27, 135,
                                    186, 166, 123, 22, 142, 235, 81, 214, 241, 20, 88, 140, 242, 240, 222, 70, 221, 204,
                                    94, 190, 15, 52, 131, 239, 20, 28],
        g2_generator_montgomery_le=[38, 32, 188, 2, 209, 181, 131, 142, 114, 1, 123, 73, 53, 25, 235, 220, 223, 26, 129,
                                    151, 71, 38, 184, 251, 59, 80, 150, 175, 65, 56, 87, 25, 64, 97, 76, 168, 125, 115,
                                    180, 175, 196, 216, 2, 88, 90, 221, 67, 96, 134, 47, 160, 82, 252, 80, 233, 9, 107,
                                    123, 234, 58, 131, 240, 254, 20, 246, 233, 107, 136, 157, 250, 157, 97, 120, 155, 158,
                                    245, 151, 210, 127, 254, 254, 125, 27, 35, 98, 26, 158, 255, 6, 66, 158, 174, 235, 126,
                                    253, 40, 238, 86, 24, 199, 86, 91, 9, 100, 187, 60, 125, 50, 34, 249, 87, 220, 118, 16,
                                    53, 51, 190, 53, 249, 85, 130, 100, 253, 147, 230, 160, 164, 13],
        g2_generator_decimal=dict(
            x_c0="10857046999023057135944570762232829481370756359578518086990519993285655852781",
            x_c1="11559732032986387107991004021392285783925812861821192530917403151452391805634",
            y_c0="8495653923123431417604973247489272438418190587263600148770280649306958101930",
            y_c1="4082367875863433681332203403145435568316851327593401208105741076214120093531"),
        fr_modulus_le_hex="010000f093f5e1439170b97948e833285d588181b64550b829a031e1724e6430",
        fr_modulus_hex="30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001",
        r1cs_sample_hex=(
            "72316373" "01000000" "03000000"
            "01000000" "40000000" "00000000"
            "20000000"
            "010000f0" "93f5e143" "9170b979" "48e83328" "5d588181" "b64550b8" "29a031e1" "724e6430"
            "07000000" "01000000" "02000000" "03000000" "e8030000" "00000000" "03000000"
            "02000000" "88020000" "00000000"
            "02000000"
            "05000000" "03000000" + "00000000" * 7 +
            "06000000" "08000000" + "00000000" * 7 +
            "03000000"
            "00000000" "02000000" + "00000000" * 7 +
            "02000000" "14000000" + "00000000" * 7 +
            "03000000" "0C000000" + "00000000" * 7 +
            "02000000"
            "00000000" "05000000" + "00000000" * 7 +
            "02000000" "07000000" + "00000000" * 7 +
            "03000000"
            "01000000" "04000000" + "00000000" * 7 +
            "04000000" "08000000" + "00000000" * 7 +
            "05000000" "03000000" + "00000000" * 7 +
            "02000000"
            "03000000" "2C000000" + "00000000" * 7 +
            "06000000" "06000000" + "00000000" * 7 +
            "00000000"
            "01000000"
            "06000000" "04000000" + "00000000" * 7 +
            "03000000"
            "00000000" "06000000" + "00000000" * 7 +
            "02000000" "0B000000" + "00000000" * 7 +
            "03000000" "05000000" + "00000000" * 7 +
            "01000000"
            "06000000" "58020000" + "00000000" * 7 +
            "03000000" "38000000" "00000000"
            "00000000" "00000000"
            "03000000" "00000000"
            "0a000000" "00000000"
            "0b000000" "00000000"
            "0c000000" "00000000"
            "0f000000" "00000000"
            "44010000" "00000000"),
        r1cs_sample_expected=dict(version=1, field_size=32, n_wires=7, n_pub_out=1, n_pub_in=2, n_prv_in=3, n_labels=0x03E8,
                                  n_constraints=3, n_constraints_len=3, c0_a_len=2, c0_a0_wire=5, c0_a0_coeff=3,
                                  c2_b0_wire=0, c2_b0_coeff=6, c1_c_len=0, wire_mapping_len=7, wire_mapping_1=3),
    )
    json.dump(kats, open(os.path.join(HERE, "reference_kats.json"), "w"), indent=1)

    # ---- Groth16: hand-checkable D = 8 circuit -------------------------------------------------------
    x, y, b = 3, 11, 1
    z = x * y % o.R
    out_ = z * z % o.R
    t = (x + 2 * y) * (3 * z) % o.R
    w = [1, out_, x, y, z, t, b]
    A = [[(1, 2)], [(1, 4)], [(1, 2), (2, 3)], [(1, 6)]]
    B = [[(1, 3)], [(1, 4)], [(3, 4)], [(1, 6), (o.R - 1, 0)]]
    Cm = [[(1, 4)], [(1, 1)], [(1, 5)], []]
    rs = [(0, 0), (rng.randrange(o.R), rng.randrange(o.R)), (0, rng.randrange(o.R)), (rng.randrange(o.R), 0)]
    case = groth16_case("d8", (A, B, Cm), 3, 4, 7, w, 1, True, rs)
    case["matrices"] = rows_json((A, B, Cm))
    json.dump(case, open(os.path.join(HERE, "groth16_d8.json"), "w"), indent=1)
    print("d8 done")

    # ---- Groth16: DummyCircuit shape of creds/src/rangeproof.rs:442-487 (m = M = 924, D = 1024) ------
    a_val, b_val = 7, rng.randrange(o.R)
    mats, l, m, M, w = o.dummy_circuit(a_val, b_val, 924, 924, 5)
    rs = [(0, 0), (rng.randrange(o.R), rng.randrange(o.R))]
    case = groth16_case("dummy1024", mats, l, m, M, w, 2, False, rs)
    case["dummy"] = dict(a=hex(a_val), b=hex(b_val), num_variables=924, num_constraints=924, num_inputs=5)
    json.dump(case, open(os.path.join(HERE, "groth16_dummy1024.json"), "w"), indent=1)
    print("dummy1024 done")

    # ---- Groth16: synthetic 'tiny' (every wire live; D = 256) ----------------------------------------
    from crescent_credentials_amd import workloads as wl
    cm, wb = wl.synthetic_circuit(0xC5E5CE47, 4, 200, 240, 0.5, 3)
    mats = wl.matrices_to_rows(cm)
    w = wl.witness_to_ints(wb)
    rs = [(0, 0), (rng.randrange(o.R), rng.randrange(o.R))]
    case = groth16_case("tiny", mats, 4, 200, 240, w, 3, False, rs)
    case["matrices"] = rows_json(mats)
    json.dump(case, open(os.path.join(HERE, "groth16_tiny.json"), "w"), indent=1)
    print("tiny done")

    # ---- NTT vectors -------------------------------------------------------------------------------
    ntt_cases = []
    for logn in (0, 1, 3, 10):
        n = 1 << logn
        v = [rng.randrange(o.R) for _ in range(n)]
        entry = dict(log_n=logn, seed_values=[hex(x) for x in v] if logn <= 3 else None, rng_seed=1000 + logn)
        if logn > 3:
            r2 = random.Random(1000 + logn)
            v = [r2.randrange(o.R) for _ in range(n)]
        outs = dict(fft=o.fft(v), ifft=o.ifft(v), coset_fft=o.coset_fft(v), coset_ifft=o.coset_ifft(v))
        if logn <= 3:
            entry["outputs"] = {k: [hex(x) for x in val] for k, val in outs.items()}
        entry["outputs_sha256"] = {k: o.sha256_hex(b"".join(o.fe_bytes(x) for x in val)) for k, val in outs.items()}
        ntt_cases.append(entry)
    json.dump(dict(omega_2_28=hex(o.FR_ROOT_2_28), generator=o.FR_GENERATOR, cases=ntt_cases),
              open(os.path.join(HERE, "ntt.json"), "w"), indent=1)
    print("ntt done")

    # ---- MSM vectors: n in {1, 2, 33, 1000}, with zero/one scalars and identity bases ---------------
    t1 = o.G1.fixed_base_table(o.G1_GEN, 8)
    t2 = o.G2.fixed_base_table(o.G2_GEN, 8)
    msm_cases = []
    for n in (1, 2, 33, 1000):
        r2 = random.Random(2000 + n)
        ks = [r2.randrange(1, o.R) for _ in range(n)]
        sc = [r2.randrange(o.R) for _ in range(n)]
        for i in range(n):     # sprinkle the edge cases the survey lists
            u = r2.random()
            if n >= 33 and u < 0.15: sc[i] = 0
            elif n >= 33 and u < 0.30: sc[i] = 1
            elif n >= 33 and u < 0.33: sc[i] = o.R - 1
            elif n >= 33 and u < 0.36: ks[i] = 0            # identity base
            elif n >= 33 and u < 0.40: ks[i] = ks[i - 1]     # repeated base (forces the doubling branch)
        if n == 2:
            ks[1] = ks[0]; sc[1] = sc[0]                     # P + P inside one bucket
        g1 = o.G1.batch_to_affine([o.G1.fixed_base_mul(t1, k) for k in ks])
        g2 = o.G2.batch_to_affine([o.G2.fixed_base_mul(t2, k) for k in ks])
        e1 = sum(k * s for k, s in zip(ks, sc)) % o.R
        r1 = o.G1.to_affine(o.G1.mul_affine(o.G1_GEN, e1))
        rr2 = o.G2.to_affine(o.G2.mul_affine(o.G2_GEN, e1))
        if n <= 33:   # cross-check the closed form against the oracle's own Pippenger and serial sum
            assert o.G1.to_affine(o.G1.msm(g1, sc)) == r1 == o.G1.to_affine(o.G1.msm_naive(g1, sc))
            assert o.G2.to_affine(o.G2.msm(g2, sc)) == rr2
        msm_cases.append(dict(n=n, base_dlogs=[hex(k) for k in ks], scalars=[hex(s) for s in sc],
                              g1_bases_sha256=o.sha256_hex(b"".join(o.g1_packed(p) for p in g1)),
                              g2_bases_sha256=o.sha256_hex(b"".join(o.g2_packed(p) for p in g2)),
                              g1_result=hx(o.g1_packed(r1)), g2_result=hx(o.g2_packed(rr2))))
    json.dump(dict(cases=msm_cases), open(os.path.join(HERE, "msm.json"), "w"), indent=1)
    print("msm done")


if __name__ == "__main__":
    main()
