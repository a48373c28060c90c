"""CPU-side checks of the C ABI: the library loads, exports every symbol include/crescent_gpu.h
declares, and its host-only entry points (the .r1cs reader) behave like the reference's
(forks/circom-compat/src/circom/r1cs_reader.rs:264-345).  No GPU compute is issued here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden

K = load_golden("reference_kats.json")


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "crescent_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(cc):
    syms = _declared_symbols()
    assert len(syms) >= 18
    L = ctypes.CDLL(cc.library_path())
    for s in syms:
        assert hasattr(L, s), "libcrescent_gpu.so does not export %s" % s
    # and the Python binding table covers the header exactly
    from crescent_credentials_amd import api
    assert sorted(api._SIGNATURES) == syms


def test_version_and_error_strings(cc):
    L = cc.lib()
    assert b"gfx950" in L.cg_version()
    rc = L.cg_r1cs_parse(None, 0, None)
    assert rc == -1 and b"null" in L.cg_last_error()


def test_r1cs_reader_kat(cc):
    f = cc.R1CSFile(bytes.fromhex(K["r1cs_sample_hex"]))
    e = K["r1cs_sample_expected"]
    h = f.header
    assert h["field_size"] == 32 and h["n_wires"] == e["n_wires"] and h["n_pub_out"] == e["n_pub_out"]
    assert h["n_pub_in"] == e["n_pub_in"] and h["n_prv_in"] == e["n_prv_in"]
    assert h["n_labels"] == e["n_labels"] and h["n_constraints"] == e["n_constraints"]
    c0, c1, c2 = f.constraint(0), f.constraint(1), f.constraint(2)
    assert len(c0[0]) == e["c0_a_len"] and c0[0][0] == (e["c0_a0_wire"], e["c0_a0_coeff"])
    assert c2[1][0] == (e["c2_b0_wire"], e["c2_b0_coeff"]) and len(c1[2]) == e["c1_c_len"]
    assert len(f.wire_mapping) == e["wire_mapping_len"] and int(f.wire_mapping[1]) == e["wire_mapping_1"]
    assert (f.num_inputs, f.num_aux, f.num_variables) == (4, 3, 7)      # r1cs_reader.rs:26-38


def test_r1cs_reader_matches_oracle(cc, oracle):
    data = bytes.fromhex(K["r1cs_sample_hex"])
    f = cc.R1CSFile(data)
    po = oracle.parse_r1cs(data)
    for i, (a, b, c) in enumerate(po["constraints"]):
        assert f.constraint(i) == (a, b, c)


@pytest.mark.parametrize("mutate,msg", [
    (lambda b: b.__setitem__(0, b[0] ^ 1), "magic"),
    (lambda b: b.__setitem__(4, 2), "version"),
    (lambda b: b.__setitem__(24, 31), "32-byte"),
    (lambda b: b.__setitem__(28, b[28] ^ 1), "bn256"),
])
def test_r1cs_reader_rejections(cc, mutate, msg):
    b = bytearray(bytes.fromhex(K["r1cs_sample_hex"]))
    mutate(b)
    with pytest.raises(cc.CrescentGpuError) as ei:
        cc.R1CSFile(bytes(b))
    assert ei.value.code == -7 and msg in str(ei.value)


def test_r1cs_truncated(cc):
    b = bytes.fromhex(K["r1cs_sample_hex"])
    with pytest.raises(cc.CrescentGpuError):
        cc.R1CSFile(b[:200])


def test_workload_generator_satisfies(oracle):
    from crescent_credentials_amd import workloads as wl
    for bf in (0.0, 0.9):
        cm, w = wl.synthetic_circuit(7, 5, 300, 330, bf, 3)
        A, B, C = wl.matrices_to_rows(cm)
        wi = wl.witness_to_ints(w)
        assert wi[0] == 1 and len(wi) == 330
        for i in range(300):
            assert oracle.evaluate_constraint(A[i], wi) * oracle.evaluate_constraint(B[i], wi) % oracle.R == oracle.evaluate_constraint(C[i], wi)
    zeros = sum(1 for x in wi if x == 0); ones = sum(1 for x in wi if x == 1)
    assert zeros + ones > 0.8 * 330


def test_no_cpu_fallback_in_product():
    """The product package must never reach into oracle/ (a CPU fallback would void parity)."""
    pkg = os.path.join(ROOT, "crescent-credentials_amd")
    for dirpath, _, files in os.walk(pkg):
        if "_build" in dirpath or "__pycache__" in dirpath:
            continue
        for fn in files:
            path = os.path.join(dirpath, fn)
            if fn.endswith(".py"):
                txt = open(path).read()
                assert not re.search(r"^\s*(import|from)\s+\S*(oracle|cpu_ref)", txt, flags=re.M), "%s imports the oracle" % fn
                assert "oracle" + os.sep not in txt, "%s references oracle/" % fn
            elif fn.endswith((".hip", ".hpp", ".cpp", ".h")):
                txt = open(path, errors="replace").read()
                assert not re.search(r'#include\s+"[^"]*oracle', txt), "%s includes oracle code" % fn


# ---- ark-serialize proving key reader / writer (host-only) --------------------------------------
def _oracle_pk(oracle, j):
    g1s = lambda h: [oracle.g1_unpack(bytes.fromhex(h)[i:i + 64]) for i in range(0, len(h) // 2, 64)]
    g2s = lambda h: [oracle.g2_unpack(bytes.fromhex(h)[i:i + 128]) for i in range(0, len(h) // 2, 128)]
    vk = dict(alpha_g1=g1s(j["alpha_g1"])[0], beta_g2=g2s(j["beta_g2"])[0], gamma_g2=g2s(j["gamma_g2"])[0],
              delta_g1=g1s(j["delta_g1"])[0], delta_g2=g2s(j["delta_g2"])[0], gamma_abc_g1=g1s(j["gamma_abc_g1"]))
    return dict(vk=vk, beta_g1=g1s(j["beta_g1"])[0], delta_g1=vk["delta_g1"], a_query=g1s(j["a_query"]),
                b_g1_query=g1s(j["b_g1_query"]), b_g2_query=g2s(j["b_g2_query"]), h_query=g1s(j["h_query"]),
                l_query=g1s(j["l_query"]))


def test_proving_key_ark_serialize_roundtrip(cc, oracle):
    """bytes written by the oracle's ark-serialize encoder (data_structures.rs field order, SWFlags) parse into the
    golden packed arrays, and serialise back to the same bytes (flags recomputed by the library)."""
    g = load_golden("groth16_d8.json")
    blob = oracle.pk_uncompressed(_oracle_pk(oracle, g["pk"]))
    # prover_params.bin carries more after the key (PreparedVerifyingKey, config string): trailing bytes are left alone
    pk, used = cc.proving_key_from_bytes(blob + b"\x07" * 100)
    assert used == len(blob)
    j = g["pk"]
    for name in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query"):
        assert getattr(pk, name).tobytes().hex() == j[name], name
    assert pk.vk.gamma_abc_g1.tobytes().hex() == j["gamma_abc_g1"]
    assert pk.vk.alpha_g1.tobytes().hex() == j["alpha_g1"] and pk.vk.gamma_g2.tobytes().hex() == j["gamma_g2"]
    assert pk.beta_g1.tobytes().hex() == j["beta_g1"] and pk.delta_g1.tobytes().hex() == j["delta_g1"]
    # the d8 key has identity points (zero QAP columns): they travel as the infinity flag, not as coordinates
    assert any(blob[i + 63] & 0x40 for i in range(0, 64, 64)) or bytes(64) in [pk.a_query[k:k + 64].tobytes() for k in range(0, pk.a_query.size, 64)]
    assert cc.proving_key_to_bytes(pk) == blob


def test_proving_key_parse_rejects_truncation(cc, oracle):
    g = load_golden("groth16_d8.json")
    blob = oracle.pk_uncompressed(_oracle_pk(oracle, g["pk"]))
    with pytest.raises(cc.CrescentGpuError) as ei:
        cc.proving_key_from_bytes(blob[:-10])
    assert ei.value.code == -7
    bad = bytearray(blob)
    off = 64 + 128 + 128 + 64 + 128          # gamma_abc_g1 length field
    bad[off:off + 8] = (1 << 40).to_bytes(8, "little")
    with pytest.raises(cc.CrescentGpuError):
        cc.proving_key_from_bytes(bytes(bad))


def test_circuit_load_validates_before_touching_the_gpu(cc):
    """argument errors are reported by the C ABI with the reference's vocabulary (no GPU needed)"""
    L = cc.lib()
    assert L.cg_circuit_load(None, None, None, 1, 1, 2, None) == -1
    from crescent_credentials_amd import api
    pk = api._CgProvingKey()
    pk.coord_form = 0
    pk.a_len = pk.b_g1_len = pk.b_g2_len = 7
    pk.l_len = 4
    pk.h_len = 5                     # must be D - 1 = 7 for m = 4, l = 3
    abc = (api._CgCsr * 3)()
    h = ctypes.c_void_p()
    rc = L.cg_circuit_load(ctypes.byref(h), ctypes.byref(pk), abc, 3, 4, 7, None)
    assert rc == -6 and b"h_query" in L.cg_last_error()
    rc = L.cg_circuit_load(ctypes.byref(h), ctypes.byref(pk), abc, 3, (1 << 28) + 5, (1 << 28) + 9, None)
    assert rc == -5            # PolynomialDegreeTooLarge (r1cs_to_qap.rs:156-157)
    # options
    pk.h_len = 7
    for kw, msg in ((dict(window_bits=1), b"window_bits"), (dict(window_bits=23), b"window_bits"), (dict(window_bits=-3), b"window_bits"),
                    (dict(shard_count=4, shard_rank=4), b"shard_rank"), (dict(shard_count=2, shard_rank=-1), b"shard_rank"),
                    (dict(proof_slots=-1), b"proof_slots"), (dict(flags=256), b"flags"), (dict(flags=1 << 20), b"flags"),
                    (dict(flags=2 | 4), b"exclusive")):
        opt = api._CgOptions(device=-1, **kw)
        rc = L.cg_circuit_load(ctypes.byref(h), ctypes.byref(pk), abc, 3, 4, 7, ctypes.byref(opt))
        assert rc == -1 and msg in L.cg_last_error(), kw
    # a struct with consistent lengths but unset pointers is an argument error, not a fault
    rc = L.cg_circuit_load(ctypes.byref(h), ctypes.byref(pk), abc, 3, 4, 7, None)
    assert rc == -1 and b"null key point" in L.cg_last_error()
    import numpy as np
    pt = np.zeros(128, np.uint8)
    for f in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2"):
        setattr(pk, f, pt.ctypes.data)
    rc = L.cg_circuit_load(ctypes.byref(h), ctypes.byref(pk), abc, 3, 4, 7, None)
    assert rc == -1 and b"null query pointer" in L.cg_last_error()
    q = np.zeros(7 * 128, np.uint8)
    for f in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query"):
        setattr(pk, f, q.ctypes.data)
    rc = L.cg_circuit_load(ctypes.byref(h), ctypes.byref(pk), abc, 3, 4, 7, None)
    assert rc == -1 and b"null row_ptr" in L.cg_last_error()
    # the witness-map handle validates the same way
    q2 = ctypes.c_void_p()
    assert L.cg_qap_load(ctypes.byref(q2), abc, 3, 4, 7, -1) == -1 and b"null row_ptr" in L.cg_last_error()
    assert L.cg_qap_load(ctypes.byref(q2), abc, 3, (1 << 28) + 5, (1 << 28) + 9, -1) == -1
    assert L.cg_qap_load(ctypes.byref(q2), abc, 0, 4, 7, -1) == -1


def _c_struct_fields(name):
    src = open(os.path.join(ROOT, "include", "crescent_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        m = re.search(r"(\w+)\s*(\[\d+\])?\s*$", decl.strip())
        if m and decl.strip():
            out.append(m.group(1))
    return out


_RUST_OF_C = {"uint64_t": "u64", "uint32_t": "u32", "int32_t": "i32", "int64_t": "i64", "uint8_t": "u8", "float": "f32", "double": "f64",
              "int": "c_int", "char": "c_char", "void": "c_void"}


def _rust_type_of(ctype: str, array: str) -> str:
    """the Rust spelling of a C declaration's type as sys.rs writes it"""
    t = " ".join(ctype.split())
    const = t.startswith("const ")
    t = t[6:] if const else t
    stars = t.count("*")
    base = _RUST_OF_C[t.replace("*", "").strip()]
    for _ in range(stars):
        base = ("*const " if const else "*mut ") + base
    return "[%s; %s]" % (base, array) if array else base


def _c_struct_field_types(name):
    src = open(os.path.join(ROOT, "include", "crescent_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        ctype, rest = re.match(r"(.*?[\s\*])(\w+(?:\s*\[\d+\])?(?:\s*,\s*\*?\s*\w+)*)$", decl).groups()
        for item in rest.split(","):             # `const uint8_t *a, *b;` declares several fields of one type
            item = item.strip()
            ptr = item.startswith("*")
            m = re.match(r"\*?\s*(\w+)\s*(?:\[(\d+)\])?$", item)
            out.append((m.group(1), _rust_type_of(ctype + ("*" if ptr else ""), m.group(2))))
    return out


def test_rust_shim_struct_field_types_match_header():
    """the repr(C) structs of sys.rs carry the header's field TYPES, not only its names and order (the crate has
    never met rustc, so the layout is checked here): u64 / i32 / f32 widths, pointer constness, array lengths"""
    rs = open(os.path.join(ROOT, "integration", "rust", "crescent-gpu", "src", "sys.rs")).read()
    for name in ("cg_proving_key", "cg_csr", "cg_options", "cg_timings", "cg_ctx_info", "cg_load_timings"):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, rs, flags=re.S).group(1)
        rust = [(f, " ".join(t.split())) for f, t in re.findall(r"pub (\w+):\s*([^,\n]+),", body)]
        assert rust == _c_struct_field_types(name), name
    # and the ctypes mirror's widths
    from crescent_credentials_amd import api
    width = {"u64": 8, "u32": 4, "i32": 4, "f32": 4}
    for cls, name in ((api._CgOptions, "cg_options"), (api.CgTimings, "cg_timings"), (api.CgCtxInfo, "cg_ctx_info"), (api._CgProvingKey, "cg_proving_key"),
                      (api.CgLoadTimings, "cg_load_timings")):
        for (fname, ctype), (hname, rtype) in zip(cls._fields_, _c_struct_field_types(name)):
            assert fname == hname
            m = re.match(r"\[(\w+); (\d+)\]", rtype)
            want = width[m.group(1)] * int(m.group(2)) if m else (8 if rtype.startswith("*") else width[rtype])
            assert ctypes.sizeof(ctype) == want, (name, fname, rtype)


def test_rust_shim_function_signatures_match_header():
    """every extern fn of sys.rs takes the header's arguments: the same count, and per argument the same shape (scalar
    width, pointer or not, pointer constness)"""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "crescent_gpu.h")).read(), flags=re.S)
    rs = open(os.path.join(ROOT, "integration", "rust", "crescent-gpu", "src", "sys.rs")).read()

    def c_shape(arg):
        arg = " ".join(re.sub(r"\[\d+\]", "*", arg).split())               # an array parameter is a pointer
        if arg in ("void", ""):
            return None
        ptr = arg.count("*")
        const = arg.startswith("const ")
        base = re.sub(r"^const ", "", arg).replace("*", " ").split()[0]
        if base == "struct":
            base = re.sub(r"^const ", "", arg).replace("*", " ").split()[1]
        return (ptr, const if ptr else False, _RUST_OF_C.get(base, base))

    def rust_shape(arg):
        t = " ".join(arg.split(":", 1)[1].split())
        ptr = t.count("*const ") + t.count("*mut ")
        const = t.startswith("*const ")
        base = t.replace("*const ", "").replace("*mut ", "")
        return (ptr, const if ptr else False, base)

    checked = 0
    for name, args in re.findall(r"pub fn (cg_[a-z0-9_]+)\s*\((.*?)\)", rs, flags=re.S):
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, hdr, flags=re.S)
        assert m, name
        c_args = [c_shape(a) for a in m.group(1).split(",")]
        c_args = [a for a in c_args if a is not None]
        r_args = [rust_shape(a) for a in args.split(",") if a.strip()]
        assert len(c_args) == len(r_args), (name, c_args, r_args)
        for ca, ra in zip(c_args, r_args):
            assert ca[0] == ra[0] and ca[1] == ra[1], (name, ca, ra)            # pointer depth and constness
            if ca[2] in _RUST_OF_C.values():                                       # scalar / byte pointers: the same width
                assert ca[2] == ra[2] or {ca[2], ra[2]} <= {"c_int", "i32"} or {ca[2], ra[2]} <= {"c_void", "u8"}, (name, ca, ra)
        checked += 1
    assert checked >= 8


def test_header_is_strict_c_and_the_c_caller_links():
    """include/crescent_gpu.h compiles as C11 with -pedantic -Werror (a C or cgo/bindgen consumer sees no C++), and
    integration/c/crescent_prove - the reference's create_client_state as a plain C program - is built by build() and
    reaches its argument check without a GPU."""
    import subprocess
    import tempfile
    inc = os.path.join(ROOT, "include")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        with open(src, "w") as f:
            f.write('#include <crescent_gpu.h>\nint main(void) { cg_options o; cg_timings t; (void)o; (void)t; return sizeof(cg_proving_key) ? 0 : 1; }\n')
        r = subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, src],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    exe = os.path.join(ROOT, "integration", "c", "crescent_prove")
    assert os.path.exists(exe), "integration/c/crescent_prove missing: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr
    r = subprocess.run([exe, "a", "b", "c", "d", "--rs", "zz", "1"], capture_output=True, text=True)
    assert r.returncode == 2 and "--rs" in r.stderr
    # r >= the scalar modulus is refused (r, s cross the ABI as canonical integers)
    r = subprocess.run([exe, "a", "b", "c", "d", "--rs", "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 2


def test_rust_shim_bindings_match_header():
    """integration/rust/crescent-gpu/src/sys.rs (compile-untested: no Rust toolchain here) declares only functions the
    header declares, and its repr(C) structs list the header's fields in the header's order."""
    rs = open(os.path.join(ROOT, "integration", "rust", "crescent-gpu", "src", "sys.rs")).read()
    fns = re.findall(r"pub fn (cg_[a-z0-9_]+)\s*\(", rs)
    assert len(fns) >= 8 and set(fns) <= set(_declared_symbols())
    for name in ("cg_proving_key", "cg_csr", "cg_options", "cg_timings", "cg_ctx_info", "cg_load_timings"):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, rs, flags=re.S).group(1)
        fields = re.findall(r"pub (\w+):", body)
        assert fields == _c_struct_fields(name), name
    # the ctypes mirrors agree too
    from crescent_credentials_amd import api
    assert [f for f, _ in api._CgOptions._fields_] == _c_struct_fields("cg_options")
    assert [f for f, _ in api.CgTimings._fields_] == _c_struct_fields("cg_timings")
    assert [f for f, _ in api._CgProvingKey._fields_] == _c_struct_fields("cg_proving_key")
    assert [f for f, _ in api.CgCtxInfo._fields_] == _c_struct_fields("cg_ctx_info")
    # the resident-matrices cache of the shim is keyed by a digest of EVERY term (round 2 sampled every 1024th) and bounded
    lib_rs = open(os.path.join(ROOT, "integration", "rust", "crescent-gpu", "src", "lib.rs")).read()
    assert "% 1024" not in lib_rs and "QAP_CACHE_MAX" in lib_rs and "cg_host_alloc" in lib_rs


def test_parsers_survive_mutated_input(cc, oracle):
    """the two parsers of untrusted bytes (.r1cs, ark-serialize proving key) either succeed or return an error on
    randomly corrupted, truncated and extended input - never crash, never hand out a view past their buffers"""
    import random
    from crescent_credentials_amd import api
    L = cc.lib()
    rng = random.Random(7)
    base = bytearray(bytes.fromhex(K["r1cs_sample_hex"]))
    parsed = 0
    for _ in range(3000):
        b = bytearray(base)
        for _ in range(rng.choice([1, 1, 2, 4, 8])):
            pos = rng.randrange(len(b))
            b[pos] = rng.choice([0, 0xFF, b[pos] ^ (1 << rng.randrange(8)), rng.randrange(256)])
        if rng.random() < 0.2:
            b = b[:rng.randrange(len(b) + 1)]
        if rng.random() < 0.05:
            b += bytes(rng.randrange(256) for _ in range(rng.randrange(40)))
        arr = (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) or b"\0")
        h = ctypes.c_void_p()
        if L.cg_r1cs_parse(arr, len(b), ctypes.byref(h)) == 0:
            parsed += 1
            hdr = api._CgR1csHeader(); abc = (api._CgCsr * 3)(); wm = ctypes.c_void_p()
            assert L.cg_r1cs_get(h, ctypes.byref(hdr), abc, ctypes.byref(wm)) == 0
            for m in abc:                                    # walk what the views claim to cover
                n = int(m.nnz)
                if n:
                    cols = np.ctypeslib.as_array(ctypes.cast(m.col, ctypes.POINTER(ctypes.c_uint32)), shape=(n,))
                    assert int(cols.max()) < hdr.n_wires
            L.cg_r1cs_free(h)
    assert parsed > 100
    mats = ([[(1, 1)], [(2, 2)]], [[(1, 2)], [(1, 1)]], [[(1, 3)], [(3, 0)]])
    pk, _ = oracle.generate_parameters(mats, 2, 2, 4, 11, 12, 13, 14)
    blob = bytearray(oracle.pk_uncompressed(pk))
    parsed = 0
    for _ in range(3000):
        b = bytearray(blob)
        for _ in range(rng.choice([1, 2, 4])):
            b[rng.randrange(len(b))] = rng.choice([0, 0xFF, rng.randrange(256)])
        if rng.random() < 0.3:
            b = b[:rng.randrange(len(b) + 1)]
        arr = (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) or b"\0")
        h = ctypes.c_void_p(); used = ctypes.c_uint64()
        if L.cg_pk_parse(arr, len(b), ctypes.byref(h), ctypes.byref(used)) == 0:
            parsed += 1
            assert used.value <= len(b)
            L.cg_pk_free(h)
    assert parsed > 100


def test_shipped_library_reads_no_environment_switch(cc):
    """the tuning / A-B / fault-injection switches live in the -DCG_TUNING build only
    (libcrescent_gpu_tuning.so); the shipped library carries no `CG_...` string at all - what a host may choose is a
    cg_options flag - and no getenv call in csrc/ escapes the CG_TUNE_ENV macro."""
    import subprocess
    out = subprocess.run(["strings", cc.library_path()], capture_output=True, text=True, check=True).stdout
    hits = [ln for ln in out.splitlines() if ln.startswith("CG_")]
    assert hits == [], hits
    csrc = os.path.join(ROOT, "crescent-credentials_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        src = open(os.path.join(csrc, f)).read()
        code = re.sub(r"//[^\n]*", "", src)
        for m in re.finditer(r"\bgetenv\s*\(", code):
            line = code[code.rfind("\n", 0, m.start()) + 1:code.find("\n", m.start())]
            if 'getenv("GPU_MAX_HW_QUEUES")' in line:
                continue      # the HIP runtime's own variable (cg_init sets its default): read once, to size the copy streams
            assert f == "common.hpp" and "#define CG_TUNE_ENV" in line, "%s reads the environment: %s" % (f, line.strip())
    # the header documents the flags that replaced the host-visible switches
    hdr = open(os.path.join(ROOT, "include", "crescent_gpu.h")).read()
    for flag in ("CG_FLAG_LATENCY_MODE", "CG_FLAG_THROUGHPUT_MODE", "CG_FLAG_SPIN_WAIT", "CG_FLAG_CONTIGUOUS_H_SHARDS"):
        assert flag in hdr


def _device_disassembly(lib_path, tmp_path):
    """gfx950 disassembly of every code object bundled in a shared library (llvm-objdump, no GPU needed)"""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    work = os.path.join(str(tmp_path), "lib.so")
    shutil.copy(lib_path, work)
    subprocess.run([objdump, "--offloading", work], capture_output=True, text=True, check=True, cwd=str(tmp_path))
    text = []
    for f in sorted(os.listdir(str(tmp_path))):
        if "gfx950" in f:
            text.append(subprocess.run([objdump, "-d", os.path.join(str(tmp_path), f)], capture_output=True, text=True, check=True).stdout)
    return "\n".join(text)


def test_plan_hand_off_drains_vector_memory_before_the_barrier(cc, tmp_path):
    """in k_part_count the no-return histogram atomics must be acknowledged (s_waitcnt vmcnt(0)) before
    the s_barrier that precedes the PLAN_DONE ticket - a workgroup-scope release alone drains only lgkmcnt.  Checked on the
    shipped ISA, for every instantiation of the kernel."""
    asm = _device_disassembly(cc.library_path(), tmp_path)
    bodies = re.findall(r"<_ZN2cg12k_part_countILi\d+EEEv[^>]*>:\n(.*?)(?=\n\n|\Z)", asm, flags=re.S)
    assert len(bodies) >= 14, len(bodies)
    for body in bodies:
        lines = [ln.split("//")[0].strip() for ln in body.splitlines()]
        ops = [ln for ln in lines if ln]
        # the returning atomic (sc0) is the PLAN_DONE ticket; the last no-return global atomic before it is the histogram add
        ticket = max(i for i, ln in enumerate(ops) if ln.startswith("global_atomic_add") and "sc0" in ln)
        hist = max(i for i, ln in enumerate(ops[:ticket]) if ln.startswith("global_atomic_add") and "sc0" not in ln)
        between = ops[hist + 1:ticket]
        barrier = next(i for i, ln in enumerate(between) if ln.startswith("s_barrier"))
        assert any(ln.startswith("s_waitcnt") and "vmcnt(0)" in ln for ln in between[:barrier]), between[:barrier + 1]


def test_level1_grouping_kernels_use_no_scratch(cc, tmp_path):
    """VERDICT r4 #1b: the level-1 grouping kernels kept their four scalars in scratch (144 B per thread, 230 MB per proof).
    The shipped code objects must declare no private segment for any k_part_* kernel, nor for the accumulation kernels."""
    import shutil
    import subprocess
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(readelf) and os.path.exists(objdump)):
        pytest.skip("llvm-readelf / llvm-objdump not available")
    work = os.path.join(str(tmp_path), "lib.so")
    shutil.copy(cc.library_path(), work)
    subprocess.run([objdump, "--offloading", work], capture_output=True, text=True, check=True, cwd=str(tmp_path))
    seen = 0
    for f in sorted(os.listdir(str(tmp_path))):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", os.path.join(str(tmp_path), f)], capture_output=True, text=True, check=True).stdout
        for blk in re.split(r"\n\s*- \.", notes):
            nm = re.search(r"\.name:\s+(\S+)", blk)
            ps = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
            if nm and ps and re.search(r"k_part_(count|place)|k_accum_affine", nm.group(1)):
                seen += 1
                assert int(ps.group(1)) == 0, "%s uses %s bytes of scratch per thread" % (nm.group(1), ps.group(1))
    assert seen >= 30, seen


def test_stream_order_audit_of_the_kernel_sources():
    """VERDICT r4 #7 (the hipMemset race of round 4 was found by luck): no call in csrc/ may run on the legacy default
    stream or be a synchronous copy / fill - the default stream does not order itself against the non-blocking streams
    every context works on.  Enforced on the sources: no hipMemcpy( / hipMemset( / hipMemcpyDtoH-style synchronous forms,
    every kernel launch names a stream, every stream is created non-blocking, and hipDeviceSynchronize appears only in the
    four *_free entry points ("stream-order audit")."""
    csrc = os.path.join(ROOT, "crescent-credentials_amd", "csrc")
    dev_syncs = []
    for f in sorted(os.listdir(csrc)):
        src = open(os.path.join(csrc, f)).read()
        code = re.sub(r"//[^\n]*", "", src)
        code = re.sub(r"/\*.*?\*/", "", code, flags=re.S)
        for bad in (r"\bhipMemcpy\s*\(", r"\bhipMemset\s*\(", r"\bhipMemcpy(DtoH|HtoD|DtoD)\s*\(", r"\bhipMemcpy2D\s*\(",
                    r"\bhipMemsetD(8|16|32)\s*\(", r"\bhipStreamCreate\s*\(", r"\bhipMemcpyToSymbol\s*\(", r"\bhipMemcpyFromSymbol\s*\("):
            m = re.search(bad, code)
            assert not m, "%s: %s" % (f, code[m.start():m.start() + 60] if m else "")
        for m in re.finditer(r"<<<(.*?)>>>", code, flags=re.S):
            parts, depth, cur = [], 0, ""
            for ch in m.group(1):            # split the launch configuration at its top-level commas
                depth += ch == "("
                depth -= ch == ")"
                if ch == "," and depth == 0:
                    parts.append(cur.strip())
                    cur = ""
                else:
                    cur += ch
            parts.append(cur.strip())
            assert len(parts) == 4 and parts[3] not in ("0", "nullptr", "NULL", "hipStreamDefault"), "%s: launch without a stream: <<<%s>>>" % (f, m.group(1)[:80])
        for m in re.finditer(r"hipStreamCreateWithFlags\s*\(([^;]*?)\)\s*\)?;", code):
            assert "hipStreamNonBlocking" in m.group(1), "%s: blocking stream" % f
        dev_syncs += [f] * len(re.findall(r"\bhipDeviceSynchronize\s*\(", code))
    assert sorted(dev_syncs) == ["prover.hip", "unit.hip", "unit.hip", "unit.hip"], dev_syncs


def test_rust_shim_prints_the_reference_phases_under_print_trace():
    """VERDICT r4 #9 / SURVEY 5 "mirror 1:1": under the `print-trace` feature GpuCircuit::create_proof hands cg_prove a
    cg_timings and prints the phase names of forks/groth16/src/prover.rs (compile-untested like the rest of the crate:
    checked against the sources by text)."""
    rs = open(os.path.join(ROOT, "integration", "rust", "crescent-gpu", "src", "lib.rs")).read()
    toml = open(os.path.join(ROOT, "integration", "rust", "crescent-gpu", "Cargo.toml")).read()
    assert re.search(r"^print-trace\s*=", toml, flags=re.M)
    assert 'cfg!(feature = "print-trace")' in rs and "sys::cg_timings::default()" in rs
    body = rs[rs.index("fn print_trace"):]
    for phase, field in (("Groth16::Prover", "total_ms"), ("R1CS to QAP witness map", "witness_map_ms"), ("Compute C", "msm_h_ms"),
                         ("Compute A", "msm_a_ms"), ("Compute B in G1", "msm_b1_ms"), ("Compute B in G2", "msm_b2_ms"), ("Finish C", "finish_ms")):
        assert phase in body and field in body, phase
    # every field the printer reads exists in the header's struct
    hdr_fields = set(_c_struct_fields("cg_timings"))
    for field in re.findall(r"tm\.(\w+)", body[:body.index("impl GpuCircuit")]):
        assert field in hdr_fields, field


def test_launch_shapes_found_in_round_5_stay():
    """Two measured launch-shape findings of round 5, guarded on the sources (profiles/r05_ac_tail_workgroups.txt,
    r05_u_copy_streams_and_queues.txt): the wave-per-unit kernels around the accumulation go out as FOUR waves per workgroup
    (one per SIMD of a CU; single-wave workgroups cost 1.8 % of the rate), and a context's upload buffers share four copy-only
    streams when the runtime's hardware queues hold them beside the proof streams (cg_init asks for twenty)."""
    csrc = os.path.join(ROOT, "crescent-credentials_amd", "csrc")
    msm = open(os.path.join(csrc, "msm.hip")).read()
    m = re.search(r"static uint32_t tail_block\(\) \{.*?\n\}", msm, flags=re.S)
    assert m and re.search(r"\? x : 256\)", m.group(0)), "tail_block() no longer defaults to 256 threads"
    for kernel in ("k_combine_wave", "k_bucket_chunks", "k_bucket_chunk_sums"):
        launches = re.findall(kernel + r"<F29T><<<(.*?)>>>", msm, flags=re.S)
        assert launches and all("tail_block<F29T>()" in l or ", tb," in l for l in launches), kernel
        assert re.search(r"__launch_bounds__\(256\) " + kernel + r"\(", msm), kernel
    prover = open(os.path.join(csrc, "prover.hip")).read()
    assert 'setenv("GPU_MAX_HW_QUEUES", "20", 0)' in prover
    assert re.search(r"hwq >= n_slots \+ 4 \? 4 :", prover), "the copy-stream rule changed"
