"""Pins oracle/keycheck.py and the cpu_ref helpers under it (ref_qap_at, ref_fr_inner, ref_fr_combine, ref_fr_powers) to the
pure-Python oracle on the committed golden circuit, so that the full-size `-m gpu` tests can rely on them.  Pure CPU."""
from types import SimpleNamespace

import numpy as np
import pytest

import keycheck
from conftest import load_golden
from test_cpu_ref import _csr, _rows, _scalars


@pytest.fixture(scope="module")
def case(oracle):
    import cpu_ref
    g = load_golden("groth16_tiny.json")
    rows = _rows(g["matrices"])
    t = {k: int(v, 16) for k, v in g["trapdoor"].items()}
    l, m, M = g["num_inputs"], g["num_constraints"], g["num_variables"]
    pk_o, qap = oracle.generate_parameters(rows, l, m, M, t["tau"], t["alpha"], t["beta"], t["delta"])
    g1s = lambda pts: np.frombuffer(b"".join(oracle.g1_packed(p) for p in pts), dtype=np.uint8).copy()
    g2s = lambda pts: np.frombuffer(b"".join(oracle.g2_packed(p) for p in pts), dtype=np.uint8).copy()
    v = pk_o["vk"]
    vk = SimpleNamespace(alpha_g1=g1s([v["alpha_g1"]]), beta_g2=g2s([v["beta_g2"]]), gamma_g2=g2s([v["gamma_g2"]]),
                         delta_g1=g1s([v["delta_g1"]]), delta_g2=g2s([v["delta_g2"]]), gamma_abc_g1=g1s(v["gamma_abc_g1"]))
    pk = SimpleNamespace(vk=vk, beta_g1=g1s([pk_o["beta_g1"]]), delta_g1=g1s([pk_o["delta_g1"]]), a_query=g1s(pk_o["a_query"]),
                         b_g1_query=g1s(pk_o["b_g1_query"]), b_g2_query=g2s(pk_o["b_g2_query"]), h_query=g1s(pk_o["h_query"]),
                         l_query=g1s(pk_o["l_query"]))
    cm = SimpleNamespace(a=_csr(rows[0]), b=_csr(rows[1]), c=_csr(rows[2]))
    trap = (t["alpha"], t["beta"], t["delta"], t["tau"])
    return SimpleNamespace(g=g, rows=rows, l=l, m=m, M=M, pk=pk, pk_o=pk_o, qap=qap, cm=cm, trap=trap, ref=cpu_ref,
                           w=_scalars([int(x, 16) for x in g["witness"]]))


def test_qap_at_equals_python_oracle(case, oracle):
    """r1cs_to_qap.rs:103-147: u, a, b, c, zt and the query scalars of generator.rs:118-128,178"""
    c = case
    a, b, cc_, zt, u = c.ref.qap_at((c.cm.a, c.cm.b, c.cm.c), c.l, c.m, c.M, c.trap[3], want_u=True)
    assert keycheck._ints(u) == oracle.evaluate_all_lagrange_coefficients(c.qap["D"], c.trap[3])
    assert keycheck._ints(a) == c.qap["a"] and keycheck._ints(b) == c.qap["b"] and keycheck._ints(cc_) == c.qap["c"]
    assert zt == c.qap["zt"]
    scal = keycheck.key_scalars(oracle, c.ref, c.cm, c.l, c.m, c.M, c.trap)
    assert keycheck._ints(scal["l"]) == c.qap["l"] and keycheck._ints(scal["gabc"]) == c.qap["gamma_abc"]
    dinv = pow(c.trap[2], oracle.R - 2, oracle.R)
    assert keycheck._ints(scal["h"]) == [c.qap["zt"] * dinv * pow(c.trap[3], i, oracle.R) % oracle.R for i in range(c.qap["D"] - 1)]
    x = [3, oracle.R - 1, 7]; y = [oracle.R - 2, 5, 0]
    assert c.ref.fr_inner(_scalars(x), _scalars(y)) == sum(p * q for p, q in zip(x, y)) % oracle.R


def test_key_check_accepts_the_oracle_key_and_refuses_a_changed_one(case, oracle):
    c = case
    scal = keycheck.check_key(oracle, c.ref, c.pk, c.cm, c.l, c.m, c.M, c.trap, stride=37)
    # one entry of one query replaced by another valid point, off the strided sample: only the random linear
    # combination over ALL entries can see it
    for name, width, i in (("a_query", 64, 5), ("h_query", 64, 101), ("b_g2_query", 128, 9), ("l_query", 64, 50)):
        good = getattr(c.pk, name)
        bad = good.copy()
        j = i + 1
        while not bad[width * j:width * j + width].any():
            j += 1
        assert (bad[width * i:width * i + width] != bad[width * j:width * j + width]).any()
        bad[width * i:width * i + width] = bad[width * j:width * j + width]
        setattr(c.pk, name, bad)
        try:
            with pytest.raises(AssertionError):
                keycheck.check_key(oracle, c.ref, c.pk, c.cm, c.l, c.m, c.M, c.trap, scal=scal, stride=37)
        finally:
            setattr(c.pk, name, good)
    # an identity where the QAP column is not zero
    good = c.pk.b_g1_query
    bad = good.copy(); k = int(np.flatnonzero(bad.reshape(-1, 64).any(axis=1))[3]); bad[64 * k:64 * k + 64] = 0
    c.pk.b_g1_query = bad
    try:
        with pytest.raises(AssertionError):
            keycheck.check_key(oracle, c.ref, c.pk, c.cm, c.l, c.m, c.M, c.trap, scal=scal, stride=37)
    finally:
        c.pk.b_g1_query = good


def test_closed_form_and_verify_on_the_golden_proofs(case, oracle):
    c = case
    scal = keycheck.key_scalars(oracle, c.ref, c.cm, c.l, c.m, c.M, c.trap)
    for p in c.g["proofs"]:
        data = bytes.fromhex(p["proof"])
        r, s = int(p["r"], 16), int(p["s"], 16)
        assert keycheck.closed_form(oracle, c.ref, scal, c.trap, r, s, c.w, c.l) == keycheck.decode_proof(oracle, data)
        assert keycheck.verify(oracle, c.pk, c.l, c.w, data)
    # a proof that is not one: C replaced by A
    assert not keycheck.verify(oracle, c.pk, c.l, c.w, data[:192] + data[:64], expect_bad_rejected=False)
