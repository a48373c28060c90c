"""BASELINE.json's configurations at FULL size on the GPU, byte-for-byte against the C restatement.

configs 2/3 : rs256 (S21, l = 20) and rs256-sd (S21, l = 26)       on one GPU
config  4   : mdl1 (S22, D = 2^22, l = 21) - one GPU here; its 8-way MSM sharding is exercised at S21 size as eight
              sharded contexts on this one GPU (cg_prove_partial x 8 + cg_assemble), which is the same code path the
              8-rank RCCL run takes with the all_gather replaced by concatenation
config  5   : rs256-db (S21, l = 28)
Every case: the circomlib-gate workload (workloads.synthetic_circuit profile "gates", ~11 terms per row) with
circom-like wires and with all-uniform wires, both key layouts (folded = default, and the reference's arrangement
CG_FLAG_H_COEFFICIENT_BASIS), r = s = 0 and random (r, s).  The CPU side (oracle/cpu_ref.c, all host cores) takes a
few seconds per proof at these sizes.
"""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0xC5E5CE47


@pytest.fixture(scope="module", autouse=True)
def _init(cc):
    rc = cc.lib().cg_init(0, None)
    assert rc == 0, cc.lib().cg_last_error()


def _threads():
    import cpu_ref
    return cpu_ref.best_threads()          # every CPU the cgroup allows (16 on the pool's boxes, whatever nproc says)


def _workload(cc, oracle, shape, bit_fraction, seed_off):
    from crescent_credentials_amd import workloads as wl
    l, m, M = wl.SHAPES[shape]
    cm, w = wl.synthetic_circuit(SEED + seed_off, l, m, M, bit_fraction, 3, profile="gates")
    rng = random.Random(SEED + seed_off)
    trap = [rng.randrange(1, oracle.R) for _ in range(4)]
    pk = cc.generate_parameters_with_qap(cm, *trap)
    return (l, m, M), cm, w, pk, rng, tuple(trap)            # trap = (alpha, beta, delta, tau)


@pytest.mark.parametrize("shape,bit_fraction", [("rs256", 0.9), ("rs256-sd", 0.9), ("rs256-sd", 0.0), ("rs256-db", 0.9),
                                                ("mdl1", 0.9), ("mdl1", 0.0)],
                         ids=lambda v: str(v))
def test_full_size_prove_equals_cpu_restatement(cc, oracle, shape, bit_fraction):
    """forks/groth16/src/prover.rs:26-136 at the sizes BASELINE.json names"""
    import cpu_ref
    import keycheck
    (l, m, M), cm, w, pk, rng, trap = _workload(cc, oracle, shape, bit_fraction, 1 + len(shape))
    nt = _threads()
    cases = [(0, 0), (rng.randrange(oracle.R), rng.randrange(oracle.R))]
    expect = [cpu_ref.prove(pk, (cm.a, cm.b, cm.c), l, m, M, w, r, s, nthreads=nt) for r, s in cases]
    # The key every proof below is made on comes out of cg_setup (the product): is it a Groth16 key for THIS circuit?
    # Fixed points, gamma_abc, a strided sample and a random linear combination over ALL entries of the five queries
    # against the trapdoor's scalars (generator.rs:118-194), none of it through csrc/setup.hip - oracle/keycheck.py.
    scal = keycheck.check_key(oracle, cpu_ref, pk, cm, l, m, M, trap, nthreads=nt)
    for (r, s), exp in zip(cases, expect):
        # the proofs are the trapdoor's closed form (no transform, no MSM) ...
        assert keycheck.closed_form(oracle, cpu_ref, scal, trap, r, s, w, l) == keycheck.decode_proof(oracle, exp), (shape, r != 0)
        # ... and the reference's acceptance criterion holds at this size (forks/groth16/src/test.rs:70-71,
        # creds/src/lib.rs:286-290, verifier.rs:44-65); a flipped public input is refused
        assert keycheck.verify(oracle, pk, l, w, exp), (shape, r != 0)
    for coefficient_basis in (False, True):
        prover = cc.Prover(pk, cm, proof_slots=2, h_coefficient_basis=coefficient_basis)
        try:
            for (r, s), exp in zip(cases, expect):
                # twice: the first proof of a context runs on the size-based windows, the second on the re-tuned ones
                assert prover.prove(w, r, s).data == exp, (shape, bit_fraction, coefficient_basis, r != 0, "first")
                assert prover.prove(w, r, s).data == exp, (shape, bit_fraction, coefficient_basis, r != 0, "retuned")
            if not coefficient_basis and shape in ("rs256-sd", "mdl1") and bit_fraction == 0.9:
                h = cpu_ref.witness_map((cm.a, cm.b, cm.c), l, m, M, w, nthreads=nt)
                assert bytes(prover.witness_map(w)) == bytes(h)
        finally:
            prover.close()


@pytest.mark.parametrize("shape,n", [("rs256-sd", 8), ("mdl1", 8), ("rs256-sd", 3)], ids=lambda v: str(v))
def test_full_size_sharded_contexts_assemble_to_the_same_proof(cc, oracle, shape, n):
    """config 4's data path at full size: n contexts, each owning 1/n of every query, produce five partial
    sums each; their concatenation (the all_gather's result) assembles to the unsharded proof's bytes.  S21 and - config
    4's own size - S22 (mdl1, D = 2^22) over 8 strided shards (Wm29Strided / k_fold29), and S21 over 3 shards (not a power
    of two: contiguous ranges of the h query's points); a satisfying and an arbitrary assignment each."""
    import cpu_ref
    import keycheck
    (l, m, M), cm, w, pk, rng, trap = _workload(cc, oracle, shape, 0.9, 77 + n)
    r, s = rng.randrange(oracle.R), rng.randrange(oracle.R)
    nt = _threads()
    # an assignment that satisfies nothing: the aux wires of the witness in another order (the prover's identities hold
    # for any assignment; the C restatement computes the reference's (ab - c)/Z quotient for it just the same)
    w2 = w.reshape(-1, 32).copy()
    w2[l:] = w2[l:][np.random.default_rng(5).permutation(M - l)]
    w2 = w2.reshape(-1)
    exp = cpu_ref.prove(pk, (cm.a, cm.b, cm.c), l, m, M, w, r, s, nthreads=nt)
    exp2 = cpu_ref.prove(pk, (cm.a, cm.b, cm.c), l, m, M, w2, r, s, nthreads=nt)
    shards = []
    try:
        for k in range(n):
            shards.append(cc.Prover(pk, cm, shard_rank=k, shard_count=n))
        parts = b"".join(p.prove_partial(w, r) for p in shards)
        got = shards[n // 2].assemble(parts, n, r, s).data
        assert got == exp
        assert keycheck.verify(oracle, pk, l, w, got)                      # verifier.rs:44-65 on the assembled proof
        parts = b"".join(p.prove_partial(w2, r) for p in shards)           # second proof of every shard: re-tuned windows
        assert shards[0].assemble(parts, n, r, s).data == exp2
        parts = b"".join(p.prove_partial(w, 0) for p in shards)            # r = 0: b1 skipped (prover.rs:102-112)
        got0 = shards[n - 1].assemble(parts, n, 0, 0).data
        # SURVEY 8e's other arrangement at the same size: ONE witness map for all shards (cg_witness_map_coset on shard 0's
        # context, the D coset values shard-major), every shard - the other n - 1 as fresh contexts that hold no witness-map
        # memory (CG_FLAG_H_SCALARS_EXTERNAL) - proves with its slice (cg_prove_partial_q): the same bytes
        import torch
        for p in shards[1:]:
            p.close()
        del shards[1:]
        for k in range(1, n):
            shards.append(cc.Prover(pk, cm, shard_rank=k, shard_count=n, h_scalars_external=True))
        D = shards[0].domain_size
        qd = torch.empty(D * 32, dtype=torch.uint8, device="cuda")
        for wit, want in ((w, exp), (w2, exp2)):
            wd = torch.from_numpy(np.ascontiguousarray(wit)).cuda()
            shards[0].witness_map_coset(wd.data_ptr(), on_device=True, out_dev=qd.data_ptr())
            slices = [shards[0].h_scalars_slice(k) for k in range(n)]
            assert sum(c for _, c in slices) == D
            parts = b"".join(p.prove_partial_q(wd.data_ptr(), qd.data_ptr() + o * 32, r, on_device=True, q_on_device=True)
                             for p, (o, c) in zip(shards, slices))
            assert shards[-1].assemble(parts, n, r, s).data == want
        del qd
    finally:
        for p in shards:
            p.close()
    whole = cc.Prover(pk, cm)
    try:
        assert got0 == whole.prove(w, 0, 0).data
    finally:
        whole.close()


def _random_canonical(n, seed):
    """n uniform field elements as canonical bytes: the top byte is kept below 0x30 (the modulus' top byte), which loses
    nothing that matters to a transform or an MSM"""
    a = np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] %= 0x30
    return a.reshape(-1)


@pytest.mark.parametrize("logn", [21, 22])
def test_full_size_transforms_equal_cpu_restatement(cc, logn):
    """cg_ntt_run at the sizes of S21 / S22 in all four modes (forward / inverse, subgroup / coset 5·<w>), element by element
    against the C restatement's radix-2 transform (r1cs_to_qap.rs:179-210) - not by round trips"""
    import cpu_ref
    nt = _threads()
    data = _random_canonical(1 << logn, 100 + logn)
    ctx = cc.NttContext(logn)
    try:
        for inverse in (False, True):
            for coset in (False, True):
                got = ctx.run(data, inverse=inverse, coset=coset)
                want = cpu_ref.ntt(data, inverse=inverse, coset=coset, nthreads=nt)
                assert np.array_equal(np.asarray(got).reshape(-1), want.reshape(-1)), (logn, inverse, coset)
    finally:
        ctx.close()


@pytest.mark.parametrize("group,logn", [(1, 21), (2, 20)], ids=lambda v: str(v))
def test_full_size_msm_equals_cpu_restatement(cc, group, logn):
    """cg_msm_run at full size - G1 over 2^21 bases, G2 over 2^20 - with uniform and with circom-like scalars (56 % zero,
    34 % one, 10 % uniform), scalars in host memory and already on the device, against the C restatement's Pippenger
    (prover.rs:66,74,266): the value itself, not a closed form"""
    import cpu_ref
    import torch
    nt = _threads()
    n = 1 << logn
    ks = _random_canonical(n, 7 + group)
    bases = np.frombuffer(cc.fixed_base_g1(ks) if group == 1 else cc.fixed_base_g2(ks), np.uint8)
    uniform = _random_canonical(n, 11 + group)
    u = np.random.default_rng(13 + group).random(n)
    circom = uniform.reshape(n, 32).copy()
    circom[u < 0.56] = 0
    ones = (u >= 0.56) & (u < 0.90)
    circom[ones] = 0
    circom[ones, 0] = 1
    circom = circom.reshape(-1)
    ref = cpu_ref.msm_g1 if group == 1 else cpu_ref.msm_g2
    ctx = cc.MsmContext(bases, group=group)
    try:
        for name, sc in (("uniform", uniform), ("circom-like", circom)):
            want = ref(bases, sc, nthreads=nt)
            assert ctx.run(sc) == want, (group, name, "host scalars")
            d = torch.from_numpy(sc).cuda()
            assert ctx.run_dev(d.data_ptr(), n) == want, (group, name, "device scalars")
            assert ctx.run(sc[:32 * (n - 12345)]) == ref(bases, sc[:32 * (n - 12345)], nthreads=nt), (group, name, "shorter scalar vector")
    finally:
        ctx.close()


def test_two_full_size_circuits_with_sixteen_slots_each_share_the_gpu(cc, oracle):
    """The reference's host serves several credential types from one process (sample/client_helper/src/main.rs:177-216):
    an rs256-sd context (S21) and an mdl1 context (S22), sixteen proof slots each, resident side by side - under 120 GB now
    that a slot holds one set of entry lists instead of five - and proving concurrently from sixteen caller threads; every
    proof is the C restatement's."""
    import cpu_ref
    from concurrent.futures import ThreadPoolExecutor
    nt = _threads()
    jobs = {}
    for shape in ("rs256-sd", "mdl1"):
        (l, m, M), cm, w, pk, rng, trap = _workload(cc, oracle, shape, 0.9, 300 + len(shape))
        r, s = rng.randrange(oracle.R), rng.randrange(oracle.R)
        jobs[shape] = dict(prover=cc.Prover(pk, cm, proof_slots=16), w=w, r=r, s=s,
                           want=cpu_ref.prove(pk, (cm.a, cm.b, cm.c), l, m, M, w, r, s, nthreads=nt))
    try:
        infos = {k: j["prover"].info() for k, j in jobs.items()}
        resident = sum(i["total_bytes"] for i in infos.values())
        assert resident < 120e9, {k: i["total_bytes"] for k, i in infos.items()}
        assert infos["rs256-sd"]["slot_bytes"] < 1.6e9                     # was 2.97 GB
        order = ["rs256-sd", "mdl1"] * 24
        with ThreadPoolExecutor(max_workers=16) as ex:
            got = list(ex.map(lambda k: jobs[k]["prover"].prove(jobs[k]["w"], jobs[k]["r"], jobs[k]["s"]).data, order))
        assert all(g == jobs[k]["want"] for g, k in zip(got, order))
    finally:
        for j in jobs.values():
            j["prover"].close()


@pytest.mark.parametrize("leave_free_gb", [4, 14], ids=lambda v: "free%dGB" % v)
def test_a_load_that_runs_out_of_device_memory_gives_everything_back(cc, oracle, leave_free_gb):
    """VERDICT r4 #6.  A host that serves several credential types (sample/client_helper/src/main.rs:177-216) is where a
    cg_circuit_load meets a full GPU: 24-30 GB per S21 context with sixteen slots.  HBM is filled (a torch allocation) until
    only `leave_free_gb` are left - 4: the load fails while the window tables are built, 14: the tables fit and the proof
    slots do not - and the load must return CG_ERR_OUT_OF_MEMORY with the failed allocation named, give back every byte
    and its stream (free memory afterwards no lower than 64 MB below what it was before), and the same circuit must load and prove correctly once
    the filler is gone.  The same for the resident-bases MSM handle (cg_msm_load_g1)."""
    import cpu_ref
    import torch
    (l, m, M), cm, w, pk, rng, trap = _workload(cc, oracle, "rs256-sd", 0.9, 500)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, total = torch.cuda.mem_get_info()
    filler_bytes = free0 - leave_free_gb * (1 << 30)
    assert filler_bytes > 0
    filler = torch.empty(filler_bytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    try:
        before, _ = torch.cuda.mem_get_info()
        assert before < (leave_free_gb + 1) * (1 << 30)
        with pytest.raises(cc.CrescentGpuError) as ei:
            cc.Prover(pk, cm, proof_slots=16)
        assert ei.value.code == -4, str(ei.value)                       # CG_ERR_OUT_OF_MEMORY
        assert "device allocation of" in str(ei.value) and "bytes failed" in str(ei.value), str(ei.value)
        after, _ = torch.cuda.mem_get_info()
        assert after > before - (64 << 20), (before, after)       # (more may be free than before: the runtime trims its pools)
        # a second failed load leaks nothing either, and the resident-bases MSM handle behaves the same
        with pytest.raises(cc.CrescentGpuError) as ei:
            cc.Prover(pk, cm, proof_slots=16)
        assert ei.value.code == -4
        if leave_free_gb <= 4:
            big = np.ascontiguousarray(pk.h_query)                       # 2^21 - 1 points x 13 window rows = 1.7 GB of tables
            small_filler = torch.empty((leave_free_gb - 1) << 30, dtype=torch.uint8, device="cuda")      # leave about 1 GB
            try:
                b2, _ = torch.cuda.mem_get_info()
                with pytest.raises(cc.CrescentGpuError) as ei:
                    cc.MsmContext(big, group=1, window_bits=20)
                assert ei.value.code == -4 and "device allocation of" in str(ei.value), str(ei.value)
                a2, _ = torch.cuda.mem_get_info()
                assert a2 > b2 - (64 << 20), (b2, a2)
            finally:
                del small_filler
        after2, _ = torch.cuda.mem_get_info()
    finally:
        del filler
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    # with the filler gone the same circuit loads, and proves what the C restatement proves
    r, s = rng.randrange(oracle.R), rng.randrange(oracle.R)
    want = cpu_ref.prove(pk, (cm.a, cm.b, cm.c), l, m, M, w, r, s, nthreads=_threads())
    prover = cc.Prover(pk, cm, proof_slots=16)
    try:
        assert prover.prove(w, r, s).data == want and prover.prove(w, r, s).data == want
    finally:
        prover.close()
    torch.cuda.synchronize()
    free_end, _ = torch.cuda.mem_get_info()
    assert free_end > free0 - (256 << 20), (free0, free_end)          # and a freed context returns its memory too
