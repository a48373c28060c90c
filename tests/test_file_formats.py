"""The cache-file formats either side of the prove step, CPU only: the product's host-side
readers/writers (csrc/serialize.hip, csrc/r1cs.hip through the C ABI; no GPU call is made) against the independent
writers/readers of oracle/ark_files.py, plus the checks that pin ark_files.py itself."""
import json
import os
import random
import struct

import numpy as np
import pytest

from conftest import load_golden


@pytest.fixture(scope="module")
def af():
    import ark_files
    return ark_files


@pytest.fixture(scope="module")
def d8(cc, oracle):
    """the hand-checkable D = 8 golden circuit: oracle-form key, product-form key, a golden proof"""
    g = load_golden("groth16_d8.json")
    a = lambda h: np.frombuffer(bytes.fromhex(h), dtype=np.uint8).copy()
    j = g["pk"]
    vk = cc.VerifyingKey(a(j["alpha_g1"]), a(j["beta_g2"]), a(j["gamma_g2"]), a(j["delta_g1"]), a(j["delta_g2"]), a(j["gamma_abc_g1"]))
    pk = cc.ProvingKey(vk, a(j["beta_g1"]), a(j["delta_g1"]), a(j["a_query"]), a(j["b_g1_query"]), a(j["b_g2_query"]),
                       a(j["h_query"]), a(j["l_query"]))
    u1 = lambda arr: [oracle.g1_unpack(bytes(arr[64 * i:64 * i + 64])) for i in range(arr.size // 64)]
    u2 = lambda arr: [oracle.g2_unpack(bytes(arr[128 * i:128 * i + 128])) for i in range(arr.size // 128)]
    ovk = dict(alpha_g1=u1(vk.alpha_g1)[0], beta_g2=u2(vk.beta_g2)[0], gamma_g2=u2(vk.gamma_g2)[0], delta_g1=u1(vk.delta_g1)[0],
               delta_g2=u2(vk.delta_g2)[0], gamma_abc_g1=u1(vk.gamma_abc_g1))
    opk = dict(vk=ovk, beta_g1=u1(pk.beta_g1)[0], delta_g1=u1(pk.delta_g1)[0], a_query=u1(pk.a_query), b_g1_query=u1(pk.b_g1_query),
               b_g2_query=u2(pk.b_g2_query), h_query=u1(pk.h_query), l_query=u1(pk.l_query))
    return dict(g=g, pk=pk, opk=opk, ovk=ovk)


def _decode_proof(oracle, data):
    b = bytearray(data)
    b[63] &= 0x3F; b[191] &= 0x3F; b[255] &= 0x3F
    return (oracle.g1_unpack(bytes(b[:64])), oracle.g2_unpack(bytes(b[64:192])), oracle.g1_unpack(bytes(b[192:])))


# ------------------------------------------------------------------------------------------------ pins of ark_files.py
def test_prepared_pairing_is_a_pairing(oracle, af):
    rng = random.Random(7)
    a, b = rng.randrange(1, oracle.R), rng.randrange(1, oracle.R)
    P = oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, a))
    Qp = oracle.G2.to_affine(oracle.G2.mul_affine(oracle.G2_GEN, b))
    e_g = af.pairing(oracle.G1_GEN, oracle.G2_GEN)
    assert e_g != oracle._f12_one()                                        # non-degenerate
    assert oracle._f12_pow(e_g, oracle.R) == oracle._f12_one()             # lands in the order-r subgroup
    assert af.pairing(P, Qp) == oracle._f12_pow(e_g, a * b % oracle.R)     # bilinear
    # 64 doubling steps + one addition step per non-zero digit below the top + the two Frobenius steps
    pq = af.g2_prepare(Qp)
    assert len(pq["ell_coeffs"]) == 64 + sum(1 for d in af.ATE_LOOP_COUNT[:-1] if d) + 2 == 91
    # the final exponentiation is the plain (q^12 - 1)/r power raised to the stated multiple
    f = af.multi_miller_loop([(P, pq)])
    assert af.final_exponentiation(f) == oracle._f12_pow(oracle.final_exponentiation(f), af.ARK_PAIRING_POWER % oracle.R)
    # identity operands drop out of the product (multi_miller_loop's filter)
    assert af.multi_miller_loop([(None, pq), (P, af.g2_prepare(None))]) == oracle._f12_one()


def test_prepared_verification_accepts_what_the_plain_check_accepts(oracle, af, d8):
    g, ovk = d8["g"], d8["ovk"]
    pvk = af.prepare_verifying_key(ovk)
    wi = [int(x, 16) for x in g["witness"]]
    pub = wi[1:g["num_inputs"]]
    for case in g["proofs"]:
        proof = _decode_proof(oracle, bytes.fromhex(case["proof"]))
        assert oracle.verify_proof(ovk, proof, pub)
        assert af.verify_with_processed_vk(pvk, pub, proof)
        bad = list(pub); bad[0] = (bad[0] + 1) % oracle.R
        assert not af.verify_with_processed_vk(pvk, bad, proof)
    # and the serialized form round-trips through the independent reader
    assert af.pvk_from_bytes(af.pvk_bytes(pvk)) == pvk


def test_client_state_size_matches_the_reference_readme(af):
    """creds/test-vectors/README.md:5-10 lists the pre-generated rs256 client_state.bin at 39K: with l = 20 public
    wires, the 36-byte prover_aux.json and a config string of a few hundred bytes, the structures restated here give
    39-40 KiB only with 91 line coefficients per prepared G2 point"""
    vk_len = 64 + 128 + 128 + 64 + 128 + 8 + 20 * 64
    pvk_len = vk_len + 384 + 2 * (8 + 91 * 192 + 1)
    size = 8 + 19 * 32 + 1 + 8 + 36 + 256 + vk_len + pvk_len + 1 + 8 + 8 + 3 + 8 + 300
    assert 39 * 1024 <= size < 40 * 1024


# ------------------------------------------------------------------------------------------------ prover_params.bin
def test_prover_params_reader_and_writer_against_the_oracle_writer(cc, oracle, af, d8):
    pk, opk, ovk = d8["pk"], d8["opk"], d8["ovk"]
    pvk = af.prepare_verifying_key(ovk)
    cfg = json.dumps({"alg": "RS256", "exp": {"type": "number", "reveal": True}, "note": "snowman ☃"})
    blob = af.prover_params_bytes(opk, pvk, cfg)
    pp = cc.ProverParams.from_bytes(blob)
    assert pp.config_str == cfg
    assert pp.groth16_pvk == af.pvk_bytes(pvk)
    assert pp.vk_bytes == oracle.vk_uncompressed(ovk)
    for name in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query", "beta_g1", "delta_g1"):
        assert bytes(getattr(pp.groth16_params, name)) == bytes(getattr(pk, name)), name
    for name in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g1", "delta_g2", "gamma_abc_g1"):
        assert bytes(getattr(pp.groth16_params.vk, name)) == bytes(getattr(pk.vk, name)), name
    assert pp.to_bytes() == blob                                     # the product's writer emits the same file
    assert cc.ProverParams(pk, pp.groth16_pvk, cfg).to_bytes() == blob
    # a reader that stops after the key still sees where the tail begins (cg_pk_parse's bytes_consumed)
    _, used = cc.proving_key_from_bytes(blob)
    assert blob[used:used + len(pp.groth16_pvk)] == pp.groth16_pvk
    # rejections: truncation anywhere in the tail, trailing bytes, a pvk blob that is not one PreparedVerifyingKey
    for cut in (used + 10, used + len(pp.groth16_pvk) - 1, len(blob) - 1):
        with pytest.raises(cc.CrescentGpuError) as ei:
            cc.ProverParams.from_bytes(blob[:cut])
        assert ei.value.code == -7
    with pytest.raises(cc.CrescentGpuError):
        cc.ProverParams.from_bytes(blob + b"\x00")
    with pytest.raises(cc.CrescentGpuError):
        cc.ProverParams(pk, pp.groth16_pvk[:-1], cfg).to_bytes()
    # a hostile vector length inside the pvk must not be trusted
    evil = bytearray(blob)
    at = used + len(pp.vk_bytes) + 384
    evil[at:at + 8] = struct.pack("<Q", 1 << 60)
    with pytest.raises(cc.CrescentGpuError):
        cc.ProverParams.from_bytes(bytes(evil))


# ------------------------------------------------------------------------------------------------ client_state.bin
def test_client_state_writer_and_reader(cc, oracle, af, d8):
    g, ovk = d8["g"], d8["ovk"]
    pvk = af.prepare_verifying_key(ovk)
    wi = [int(x, 16) for x in g["witness"]]
    pub = wi[1:g["num_inputs"]]
    proof_bytes = bytes.fromhex(g["proofs"][1]["proof"])
    cfg = '{"alg": "RS256"}'
    for aux, credtype in ((None, "jwt"), ('{"kid": "abc"}', "mdl")):
        cs = cc.ClientState.new(pub, aux, cc.Proof(proof_bytes), oracle.vk_uncompressed(ovk), af.pvk_bytes(pvk), cfg)
        cs.credtype = credtype
        blob = cs.to_bytes()
        assert blob == af.client_state_bytes(pub, aux, _decode_proof(oracle, proof_bytes), ovk, pvk, cfg, credtype)
        back = cc.ClientState.from_bytes(blob)
        assert back == cs
        parsed = af.client_state_from_bytes(blob)                     # what `show` would start from
        assert parsed["inputs"] == pub and parsed["aux"] == aux and parsed["credtype"] == credtype
        assert af.verify_with_processed_vk(parsed["pvk"], parsed["inputs"], parsed["proof"])
    # a state that `show` has already touched: randomness set, one Pedersen opening (creds/src/dlog.rs:24-29)
    g1 = oracle.g1_uncompressed(oracle.G1_GEN)
    opening = struct.pack("<Q", 2) + g1 + g1 + oracle.fe_bytes(5) + oracle.fe_bytes(6) + g1
    cs2 = cc.ClientState(pub, None, cc.Proof(proof_bytes), oracle.vk_uncompressed(ovk), af.pvk_bytes(pvk), cfg, "jwt", 12345, opening, 1)
    blob2 = cs2.to_bytes()
    assert cc.ClientState.from_bytes(blob2) == cs2
    p2 = af.client_state_from_bytes(blob2)
    assert p2["input_com_randomness"] == 12345 and p2["committed_input_openings"][0]["m"] == 5
    # rejections
    with pytest.raises(cc.CrescentGpuError):
        cc.ClientState.from_bytes(blob2[:-1])
    with pytest.raises(cc.CrescentGpuError):
        cc.ClientState.from_bytes(blob2 + b"\x01")
    with pytest.raises(cc.CrescentGpuError):                          # opening bytes that do not hold n_openings items
        cc.ClientState(pub, None, cc.Proof(proof_bytes), oracle.vk_uncompressed(ovk), af.pvk_bytes(pvk), cfg, "jwt", None, opening, 2).to_bytes()
    with pytest.raises(cc.CrescentGpuError):                          # an input that is not a field element
        _serialize_raw_input(cc, oracle, af, d8)
    bad_tag = bytearray(blob2)
    bad_tag[8 + 32 * len(pub)] = 7                                    # Option tag of `aux`
    with pytest.raises(cc.CrescentGpuError):
        cc.ClientState.from_bytes(bytes(bad_tag))


def _serialize_raw_input(cc, oracle, af, d8):
    """cg_client_state_serialize with an input >= r, through the C struct directly (the Python wrapper reduces mod r)"""
    import ctypes as C
    from crescent_credentials_amd import api
    ovk = d8["ovk"]
    vkb = np.frombuffer(oracle.vk_uncompressed(ovk), np.uint8).copy()
    pvkb = np.frombuffer(af.pvk_bytes(af.prepare_verifying_key(ovk)), np.uint8).copy()
    inp = np.frombuffer(oracle.R.to_bytes(32, "little"), np.uint8).copy()
    proof = np.zeros(256, np.uint8)
    v = api._CgClientStateView()
    v.inputs = inp.ctypes.data; v.n_inputs = 1
    v.proof = proof.ctypes.data
    v.vk_bytes = vkb.ctypes.data; v.vk_len = vkb.size
    v.pvk_bytes = pvkb.ctypes.data; v.pvk_len = pvkb.size
    out = np.zeros(int(cc.lib().cg_client_state_serialized_size(C.byref(v))), np.uint8)
    api._check(cc.lib().cg_client_state_serialize(C.byref(v), out.ctypes.data, out.size))


# ------------------------------------------------------------------------------------------------ io_locations.sym
def test_io_locations(cc, af):
    text = af.io_locations_sym({"exp_value": 3, "email_value": 4, "modulus[0]": 5, "modulus[1]": 6, "pubkey_x[0]": 9})
    io = cc.IOLocations(text)
    assert io.get_io_location("exp_value") == 3
    assert io.get_all_names() == sorted(["exp_value", "email_value", "modulus[0]", "modulus[1]", "pubkey_x[0]"])   # BTreeMap order
    assert io.get_public_key_indices() == [4, 5, 8]                   # structs.rs:80-90: location - 1, sorted
    with pytest.raises(KeyError):
        io.get_io_location("nope")
    for bad in ("a,b,c\n", "justname\n", "name,notanumber\n"):
        with pytest.raises(ValueError):
            cc.IOLocations(bad)
    assert cc.IOLocations("").get_all_names() == []


# ------------------------------------------------------------------------------------------------ .r1cs
def test_r1cs_writer_and_product_reader_agree(cc, oracle, af):
    from crescent_credentials_amd import workloads as wl
    l, m, M = 6, 300, 340
    cm, _ = wl.synthetic_circuit(3, l, m, M, 0.8, 3, profile="gates")
    rows = wl.matrices_to_rows(cm)
    blob = af.r1cs_file_bytes(rows, M, 2, l - 3, M - l)
    f = cc.R1CSFile(blob)
    assert (f.num_inputs, f.num_variables, f.num_aux) == (l, M, M - l)
    assert f.header["n_constraints"] == m and f.header["n_pub_out"] == 2 and f.header["n_pub_in"] == l - 3
    assert wl.matrices_to_rows(f.matrices) == rows
    assert list(f.wire_mapping) == list(range(M))
    # the Python oracle's reader reads the same file the same way
    parsed = oracle.parse_r1cs(blob)
    assert oracle.r1cs_to_matrices(parsed)[0] == rows


# ------------------------------------------------------------- the compiled writers / readers of oracle/cpu_ref.c (full-size files)
def test_compiled_file_writers_agree_with_the_python_oracle_and_the_product(cc, oracle, af, d8):
    """oracle/cpu_ref.c's last section writes the 0.6 GB main_c.r1cs / prover_params.bin of the full-size tests and reads them
    back for the CPU side of bench.py's cold start.  On a small circuit: its bytes are the pure-Python oracle's, its reader
    returns what was written, and the product's (multi-threaded) parsers return the same arrays."""
    import cpu_ref
    from crescent_credentials_amd import workloads as wl
    l, m, M = 5, 700, 760
    cm, w = wl.synthetic_circuit(991, l, m, M, 0.8, 3, profile="gates")
    rows = wl.matrices_to_rows(cm)
    blob = cpu_ref.write_r1cs((cm.a, cm.b, cm.c), m, M, 1, l - 2, M - l)
    assert blob.tobytes() == af.r1cs_file_bytes(rows, M, 1, l - 2, M - l)
    hdr, mats = cpu_ref.read_r1cs(blob)
    assert (hdr["num_inputs"], hdr["n_constraints"], hdr["num_variables"]) == (l, m, M)
    got = cc.R1CSFile(blob.tobytes()).matrices
    for ref, mine, prod in zip((cm.a, cm.b, cm.c), mats, (got.a, got.b, got.c)):
        for f in ("row_ptr", "col", "coeff"):
            assert np.array_equal(getattr(ref, f), getattr(mine, f)) and np.array_equal(getattr(ref, f), getattr(prod, f)), f
    # a non-canonical coefficient is refused by both readers
    bad = blob.copy()
    at = int(hdr_offset_of_first_coefficient(blob))
    bad[at:at + 32] = np.frombuffer(oracle.FR_MODULUS_LE, np.uint8)
    with pytest.raises(AssertionError):
        cpu_ref.read_r1cs(bad)
    with pytest.raises(cc.CrescentGpuError):
        cc.R1CSFile(bad.tobytes())
    # the key: the D = 8 golden key (identities and both y signs occur in it)
    pk = d8["pk"]
    ser = cpu_ref.write_pk(pk, nthreads=3)
    assert ser.tobytes() == oracle.pk_uncompressed(d8["opk"]) == cc.proving_key_to_bytes(pk)
    back, used = cpu_ref.read_pk(ser)
    assert used == ser.size
    mine, used2 = cc.proving_key_from_bytes(ser.tobytes())
    assert used2 == ser.size
    for f in ("beta_g1", "delta_g1", "a_query", "b_g1_query", "b_g2_query", "h_query", "l_query"):
        assert np.array_equal(getattr(back, f), getattr(pk, f)) and np.array_equal(getattr(mine, f), getattr(pk, f)), f
    for f in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g1", "delta_g2", "gamma_abc_g1"):
        assert np.array_equal(getattr(back.vk, f), getattr(pk.vk, f)), f


def hdr_offset_of_first_coefficient(blob) -> int:
    """byte offset of the first coefficient of the constraint section of an .r1cs whose sections are header, constraints, map"""
    off = 12                     # magic, version, section count
    for _ in range(3):
        ty, sz = struct.unpack_from("<IQ", blob, off)
        off += 12
        if ty == 2:
            p = off
            while True:          # the first block that has a term
                n = struct.unpack_from("<I", blob, p)[0]
                if n:
                    return p + 4 + 4
                p += 4
        off += sz
    raise AssertionError("no constraint section")


def test_r1cs_parser_ranges_and_threads(cc, af):
    """the two-pass parser cuts the constraints into one range per host thread: circuits smaller and larger than a range,
    empty blocks, and a header that promises more constraints than the section holds"""
    import cpu_ref
    from crescent_credentials_amd import workloads as wl
    for (l, m, M) in ((3, 1, 8), (4, 4097, 4200), (6, 20_000, 20_100)):
        cm, _ = wl.synthetic_circuit(7 + m, l, m, M, 0.5, 3, profile="gates" if m > 1 else "r1")
        blob = cpu_ref.write_r1cs((cm.a, cm.b, cm.c), m, M, 1, l - 2, M - l)
        got = cc.R1CSFile(blob.tobytes()).matrices
        for ref, prod in zip((cm.a, cm.b, cm.c), (got.a, got.b, got.c)):
            assert np.array_equal(ref.row_ptr, prod.row_ptr) and np.array_equal(ref.col, prod.col) and np.array_equal(ref.coeff, prod.coeff)
    # n_constraints patched to 2^31: refused before anything is sized by it
    hdr_at = 12 + 12
    bad = bytearray(blob.tobytes())
    struct.pack_into("<I", bad, hdr_at + 60, 1 << 31)
    with pytest.raises(cc.CrescentGpuError) as ei:
        cc.R1CSFile(bytes(bad))
    assert "unexpected end" in str(ei.value)
