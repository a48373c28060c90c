"""The oracle against the committed golden vectors (tests/golden/*.json, made by make_golden.py)."""
import random

import pytest

from conftest import load_golden


def _rows(j):
    return tuple([[(int(c, 16), col) for c, col in row] for row in mat] for mat in j)


def _unpack_pk(oracle, j):
    g1s = lambda h: [oracle.g1_unpack(bytes.fromhex(h)[i:i + 64]) for i in range(0, len(h) // 2, 64)]
    g2s = lambda h: [oracle.g2_unpack(bytes.fromhex(h)[i:i + 128]) for i in range(0, len(h) // 2, 128)]
    vk = dict(alpha_g1=g1s(j["alpha_g1"])[0], beta_g2=g2s(j["beta_g2"])[0], gamma_g2=g2s(j["gamma_g2"])[0],
              delta_g1=g1s(j["delta_g1"])[0], delta_g2=g2s(j["delta_g2"])[0], gamma_abc_g1=g1s(j["gamma_abc_g1"]))
    return dict(vk=vk, beta_g1=g1s(j["beta_g1"])[0], delta_g1=vk["delta_g1"], a_query=g1s(j["a_query"]),
                b_g1_query=g1s(j["b_g1_query"]), b_g2_query=g2s(j["b_g2_query"]), h_query=g1s(j["h_query"]),
                l_query=g1s(j["l_query"]))


def test_d8_prove_verify_and_bytes(oracle):
    g = load_golden("groth16_d8.json")
    mats = _rows(g["matrices"])
    l, m, M = g["num_inputs"], g["num_constraints"], g["num_variables"]
    w = [int(x, 16) for x in g["witness"]]
    pk = _unpack_pk(oracle, g["pk"])
    assert len(pk["a_query"]) == M and len(pk["l_query"]) == M - l and len(pk["h_query"]) == g["domain_size"] - 1
    h = oracle.witness_map_from_matrices(mats, l, m, w)
    assert [hex(x) for x in h] == g["h"]
    for case in g["proofs"]:
        r, s = int(case["r"], 16), int(case["s"], 16)
        pr = oracle.create_proof_with_reduction_and_matrices(pk, r, s, mats, l, m, w)
        assert oracle.proof_uncompressed(pr).hex() == case["proof"]
        assert oracle.proof_compressed(pr).hex() == case["proof_compressed"]
    # acceptance criterion of the reference: verifier.rs:44-77
    assert oracle.verify_proof(pk["vk"], pr, w[1:l])
    bad = list(w[1:l]); bad[0] = (bad[0] + 1) % oracle.R
    assert not oracle.verify_proof(pk["vk"], pr, bad)


def test_d8_setup_matches_golden_pk(oracle):
    g = load_golden("groth16_d8.json")
    t = g["trapdoor"]
    pk, _ = oracle.generate_parameters(_rows(g["matrices"]), g["num_inputs"], g["num_constraints"], g["num_variables"],
                                       int(t["tau"], 16), int(t["alpha"], 16), int(t["beta"], 16), int(t["delta"], 16))
    packed = b"".join(oracle.g1_packed(p) for p in pk["a_query"])
    assert packed.hex() == g["pk"]["a_query"]
    assert b"".join(oracle.g2_packed(p) for p in pk["b_g2_query"]).hex() == g["pk"]["b_g2_query"]
    assert b"".join(oracle.g1_packed(p) for p in pk["h_query"]).hex() == g["pk"]["h_query"]


def test_tiny_witness_map_hash(oracle):
    g = load_golden("groth16_tiny.json")
    w = [int(x, 16) for x in g["witness"]]
    h = oracle.witness_map_from_matrices(_rows(g["matrices"]), g["num_inputs"], g["num_constraints"], w)
    assert len(h) == g["domain_size"] == 256 and h[-1] == 0
    assert oracle.sha256_hex(b"".join(oracle.fe_bytes(x) for x in h)) == g["h_sha256"]


def test_dummy1024_shape(oracle):
    # creds/src/rangeproof.rs:442-487: m = M = 924, 5 public inputs -> ℓ = 6, D = 1024
    g = load_golden("groth16_dummy1024.json")
    d = g["dummy"]
    mats, l, m, M, w = oracle.dummy_circuit(int(d["a"], 16), int(d["b"], 16), d["num_variables"], d["num_constraints"], d["num_inputs"])
    assert (l, m, M) == (6, 924, 925) == (g["num_inputs"], g["num_constraints"], g["num_variables"])
    assert [hex(x) for x in w] == g["witness"]
    assert oracle.domain_size_for(m + l) == 1024 == g["domain_size"]
    h = oracle.witness_map_from_matrices(mats, l, m, w)
    assert oracle.sha256_hex(b"".join(oracle.fe_bytes(x) for x in h)) == g["h_sha256"]


def test_ntt_vectors(oracle):
    g = load_golden("ntt.json")
    assert int(g["omega_2_28"], 16) == oracle.FR_ROOT_2_28 and g["generator"] == oracle.FR_GENERATOR
    for c in g["cases"]:
        n = 1 << c["log_n"]
        if c["seed_values"] is not None:
            v = [int(x, 16) for x in c["seed_values"]]
        else:
            r = random.Random(c["rng_seed"])
            v = [r.randrange(oracle.R) for _ in range(n)]
        outs = dict(fft=oracle.fft(v), ifft=oracle.ifft(v), coset_fft=oracle.coset_fft(v), coset_ifft=oracle.coset_ifft(v))
        for k, val in outs.items():
            assert oracle.sha256_hex(b"".join(oracle.fe_bytes(x) for x in val)) == c["outputs_sha256"][k]
            if "outputs" in c:
                assert [hex(x) for x in val] == c["outputs"][k]
        # definition check on one output (independent of the butterfly code)
        if n >= 2:
            w = oracle.root_of_unity(n)
            assert outs["fft"][1] == sum(v[j] * pow(w, j, oracle.R) for j in range(n)) % oracle.R


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_msm_vectors_small(oracle, idx):
    c = load_golden("msm.json")["cases"][idx]
    ks = [int(x, 16) for x in c["base_dlogs"]]
    sc = [int(x, 16) for x in c["scalars"]]
    g1 = [oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, k)) for k in ks]
    assert oracle.sha256_hex(b"".join(oracle.g1_packed(p) for p in g1)) == c["g1_bases_sha256"]
    assert oracle.g1_packed(oracle.G1.to_affine(oracle.G1.msm(g1, sc))).hex() == c["g1_result"]
    if c["n"] <= 2:
        g2 = [oracle.G2.to_affine(oracle.G2.mul_affine(oracle.G2_GEN, k)) for k in ks]
        assert oracle.g2_packed(oracle.G2.to_affine(oracle.G2.msm(g2, sc))).hex() == c["g2_result"]
