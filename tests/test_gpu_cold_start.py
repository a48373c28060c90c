"""The reference's own call at the reference's artefact size, and the staged load behind it.

`create_client_state` (creds/src/lib.rs:255-301; reached from creds/src/main.rs:96-120 and, per credential, from
sample/client_helper/src/main.rs:177-216) reads main_c.r1cs (595 MB for rs256, creds/test-vectors/README.md:5-10) and
prover_params.bin (580 MB), proves ONCE and writes client_state.bin.  Here that chain runs through the compiled C caller
(integration/c/crescent_prove: nothing but include/crescent_gpu.h) on cache directories of the BASELINE shapes at full size;
the files are written by the oracle side (oracle/cpu_ref.c's writers, pinned against the pure-Python ones in
tests/test_file_formats.py), read back by the oracle's own readers for the CPU proof the bytes are compared with, and the
client state is parsed by the independent Python reader and verified with the prepared key.

The load behind it is staged (CG_FLAG_STAGED_LOAD): first proofs in the warm-up arrangement (the reference's seven
transforms, row-0 tables, one bucket set per window), the final arrangement swapped in by the library's worker.  The bytes
must not depend on which arrangement made them, under concurrent callers, across the swap.
"""
import json
import os
import random
import subprocess
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0xC5E5CE47


@pytest.fixture(scope="module")
def af():
    import ark_files
    return ark_files


@pytest.fixture(scope="module", autouse=True)
def _init(cc):
    rc = cc.lib().cg_init(0, None)
    assert rc == 0, cc.lib().cg_last_error()


def _threads():
    import cpu_ref
    return cpu_ref.best_threads()


def _oracle_vk(oracle, vk):
    g1 = lambda a: oracle.g1_unpack(bytes(a))
    g2 = lambda a: oracle.g2_unpack(bytes(a))
    n = vk.gamma_abc_g1.size // 64
    return dict(alpha_g1=g1(vk.alpha_g1), beta_g2=g2(vk.beta_g2), gamma_g2=g2(vk.gamma_g2), delta_g1=g1(vk.delta_g1),
                delta_g2=g2(vk.delta_g2), gamma_abc_g1=[g1(vk.gamma_abc_g1[64 * i:64 * i + 64]) for i in range(n)])


def make_cache_dir(cc, oracle, af, path, shape, bit_fraction=0.9, seed_off=0):
    """a Crescent cache directory for one credential type at the size of `shape`: main_c.r1cs, prover_params.bin, and the
    witness the WASM calculator would have produced.  -> dict(shape, pvk, config, files...)"""
    import cpu_ref
    from crescent_credentials_amd import workloads as wl
    l, m, M = wl.SHAPES[shape] if isinstance(shape, str) else shape
    cm, w = wl.synthetic_circuit(SEED + 40 + seed_off, l, m, M, bit_fraction, 3, profile="gates")
    rng = random.Random(SEED + 41 + seed_off)
    trap = [rng.randrange(1, oracle.R) for _ in range(4)]
    pk = cc.generate_parameters_with_qap(cm, *trap)                  # zksetup on the GPU (creds/src/lib.rs:213-252)
    ovk = _oracle_vk(oracle, pk.vk)
    pvk = af.prepare_verifying_key(ovk)                              # Groth16::process_vk (creds/src/lib.rs:232)
    config = '{"alg": "RS256", "exp": {"type": "number", "reveal": true, "max_claim_byte_len": 31}}'
    cfg = config.encode()
    files = {k: os.path.join(str(path), k) for k in ("main_c.r1cs", "prover_params.bin", "witness.bin", "client_state.bin")}
    cpu_ref.write_r1cs((cm.a, cm.b, cm.c), m, M, 2, l - 3, M - l).tofile(files["main_c.r1cs"])
    with open(files["prover_params.bin"], "wb") as f:                # ProverParams = key ‖ prepared key ‖ config (lib.rs:58-63)
        cpu_ref.write_pk(pk, nthreads=_threads()).tofile(f)
        f.write(af.pvk_bytes(pvk))
        f.write(len(cfg).to_bytes(8, "little") + cfg)
    w.tofile(files["witness.bin"])
    return dict(shape=(l, m, M), w=w, pvk=pvk, ovk=ovk, config=config, files=files, pk=pk, cm=cm)


def cpu_proof_from_files(files, shape, r, s):
    """the CPU chain on the same files: the oracle's own readers, then the C restatement's prover"""
    import cpu_ref
    l, m, M = shape
    hdr, mats = cpu_ref.read_r1cs(np.fromfile(files["main_c.r1cs"], np.uint8))
    assert (hdr["num_inputs"], hdr["n_constraints"], hdr["num_variables"]) == (l, m, M)
    pk, _ = cpu_ref.read_pk(np.fromfile(files["prover_params.bin"], np.uint8), nthreads=_threads())
    w = np.fromfile(files["witness.bin"], np.uint8)
    return cpu_ref.prove(pk, mats, l, m, M, w, r, s, nthreads=_threads())


def run_c_caller(files, r, s, extra=()):
    from conftest import ROOT
    exe = os.path.join(ROOT, "integration", "c", "crescent_prove")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    cmd = [exe, files["main_c.r1cs"], files["prover_params.bin"], files["witness.bin"], files["client_state.bin"],
           "--rs", "%x" % r, "%x" % s, "--timings-json", *extra]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    return json.loads(run.stdout.strip().splitlines()[-1]), run.stderr


@pytest.mark.parametrize("shape", ["rs256", "rs256-sd", "mdl1"])
def test_files_to_client_state_at_full_size(cc, oracle, af, tmp_path, shape):
    """main_c.r1cs (0.6 / 1.2 GB) + prover_params.bin -> client_state.bin through the compiled C caller, staged load: the
    client state parses with the independent reader, carries the file's vk / pvk / config, verifies with the prepared key,
    and its proof is the C restatement's on the same files for the same (r, s) - the FIRST proof of the context, made in
    the warm-up arrangement, and the second, made after the swap (the caller compares the two itself)."""
    cd = make_cache_dir(cc, oracle, af, tmp_path, shape, seed_off=len(shape))
    l, m, M = cd["shape"]
    rng = random.Random(5 + len(shape))
    r, s = rng.randrange(oracle.R), rng.randrange(oracle.R)
    want = cpu_proof_from_files(cd["files"], cd["shape"], r, s)
    t, log = run_c_caller(cd["files"], r, s)
    parsed = af.client_state_from_bytes(open(cd["files"]["client_state.bin"], "rb").read())
    wi = [int.from_bytes(cd["w"][32 * i:32 * i + 32].tobytes(), "little") for i in range(l)]
    assert parsed["inputs"] == wi[1:l] and parsed["config_str"] == cd["config"] and parsed["vk"] == cd["ovk"] and parsed["pvk"] == cd["pvk"]
    assert af.verify_with_processed_vk(parsed["pvk"], parsed["inputs"], parsed["proof"])          # lib.rs:288-290
    assert cc.ClientState.from_bytes(open(cd["files"]["client_state.bin"], "rb").read()).proof.data == want
    # the run's own account of itself
    assert t["staged"] == 1 and t["background"]["ready_rc"] == 0 and t["second_proof_bytes_identical"] is True, (t, log[-2000:])
    assert t["r1cs_bytes"] == os.path.getsize(cd["files"]["main_c.r1cs"]) > 36 * 16_000_000
    print("\n[cold start %s] %s" % (shape, json.dumps(t)))
    # and with a synchronous load: the same bytes
    t2, _ = run_c_caller(cd["files"], r, s, extra=("--sync-load",))
    assert t2["staged"] == 0
    assert cc.ClientState.from_bytes(open(cd["files"]["client_state.bin"], "rb").read()).proof.data == want


def _expected(cc, oracle, pk, cm, shape, w, cases):
    import cpu_ref
    l, m, M = shape
    return [cpu_ref.prove(pk, (cm.a, cm.b, cm.c), l, m, M, w, r, s, nthreads=_threads()) for r, s in cases]


@pytest.mark.parametrize("shape_name,slots", [("medium", 1), ("medium", 4), ("small", 2)])
def test_staged_load_gives_the_same_bytes_before_and_after_the_swap(cc, oracle, shape_name, slots):
    from crescent_credentials_amd import workloads as wl
    shape = wl.SHAPES[shape_name]
    l, m, M = shape
    cm, w = wl.synthetic_circuit(SEED + 7 + slots, l, m, M, 0.85, 3, profile="gates")
    rng = random.Random(31 + slots)
    pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, oracle.R) for _ in range(4)])
    cases = [(0, 0), (rng.randrange(oracle.R), rng.randrange(oracle.R))]
    exp = _expected(cc, oracle, pk, cm, shape, w, cases)
    p = cc.Prover(pk, cm, proof_slots=slots, staged_load=True)
    try:
        first = [p.prove(w, r, s).data for r, s in cases]                 # whatever arrangement is in force
        assert first == exp
        assert p.wait_ready(120_000)
        info, lt = p.info(), p.load_timings()
        assert info["warmup"] == 0 and lt["staged"] == 1 and lt["ready"] == 1 and lt["background_status"] == 0
        assert info["proof_slots"] == slots
        assert lt["fold_ms"] > 0 and lt["window_tables_ms"] > 0 and lt["ready_after_ms"] >= lt["total_ms"]
        assert [p.prove(w, r, s).data for r, s in cases] == exp           # the final arrangement
        assert [p.prove(w, r, s).data for r, s in cases] == exp           # ... and after a re-tune, if one followed
        # the witness map entry point is the reference's result in either arrangement
        import cpu_ref
        assert bytes(p.witness_map(w)) == bytes(cpu_ref.witness_map((cm.a, cm.b, cm.c), l, m, M, w, nthreads=_threads()))
    finally:
        p.close()
    # a synchronous load reports itself as such
    q = cc.Prover(pk, cm, proof_slots=slots)
    try:
        lt = q.load_timings()
        assert lt["staged"] == 0 and lt["ready"] == 1 and q.wait_ready(0) and q.info()["warmup"] == 0
        assert q.prove(w, *cases[1]).data == exp[1]
    finally:
        q.close()


def test_staged_load_is_ignored_where_it_does_not_apply(cc, oracle):
    """sharded contexts and CG_FLAG_H_COEFFICIENT_BASIS load synchronously, flag or no flag"""
    from crescent_credentials_amd import workloads as wl
    l, m, M = wl.SHAPES["small"]
    cm, w = wl.synthetic_circuit(SEED + 3, l, m, M, 0.85, 3, profile="gates")
    rng = random.Random(2)
    pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, oracle.R) for _ in range(4)])
    r, s = rng.randrange(oracle.R), rng.randrange(oracle.R)
    exp = _expected(cc, oracle, pk, cm, (l, m, M), w, [(r, s)])[0]
    a = cc.Prover(pk, cm, h_coefficient_basis=True, staged_load=True)
    shards = [cc.Prover(pk, cm, shard_rank=k, shard_count=2, staged_load=True) for k in range(2)]
    try:
        assert a.load_timings()["staged"] == 0 and a.prove(w, r, s).data == exp
        assert all(x.load_timings()["staged"] == 0 for x in shards)
        parts = b"".join(x.prove_partial(w, r) for x in shards)
        assert shards[0].assemble(parts, 2, r, s).data == exp
    finally:
        a.close()
        for x in shards:
            x.close()


def test_a_staged_context_can_be_freed_while_its_worker_runs(cc, oracle):
    """cg_circuit_free stops the worker at its next step and waits for it; the GPU is usable afterwards"""
    from crescent_credentials_amd import workloads as wl
    l, m, M = wl.SHAPES["medium"]
    cm, w = wl.synthetic_circuit(SEED + 9, l, m, M, 0.85, 3, profile="gates")
    rng = random.Random(3)
    pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, oracle.R) for _ in range(4)])
    r, s = rng.randrange(oracle.R), rng.randrange(oracle.R)
    exp = _expected(cc, oracle, pk, cm, (l, m, M), w, [(r, s)])[0]
    for delay in (0.0, 0.01, 0.05, 0.2):
        p = cc.Prover(pk, cm, proof_slots=2, staged_load=True)
        time.sleep(delay)
        p.close()
    p = cc.Prover(pk, cm, proof_slots=2, staged_load=True)
    try:
        assert p.prove(w, r, s).data == exp
        p.close()                                # right behind a warm-up proof
        p = cc.Prover(pk, cm, proof_slots=2, staged_load=True)
        assert p.prove(w, r, s).data == exp and p.wait_ready(120_000) and p.prove(w, r, s).data == exp
    finally:
        p.close()


def test_four_threads_prove_continuously_across_the_swap_at_full_size(cc, oracle):
    """rs256-sd at S21, sixteen slots, staged: four caller threads prove without pause from the moment the load returns
    until well after the final arrangement is in force; every single proof is the C restatement's, proofs were made on both
    sides of the swap, and the context ends up tuned (from a warm-up proof's statistics, or by the usual re-tune)."""
    from crescent_credentials_amd import workloads as wl
    shape = wl.SHAPES["rs256-sd"]
    l, m, M = shape
    cm, w = wl.synthetic_circuit(SEED + 12, l, m, M, 0.9, 3, profile="gates")
    rng = random.Random(77)
    pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, oracle.R) for _ in range(4)])
    cases = [(0, 0)] + [(rng.randrange(oracle.R), rng.randrange(oracle.R)) for _ in range(2)]
    exp = _expected(cc, oracle, pk, cm, shape, w, cases)
    t0 = time.perf_counter()
    p = cc.Prover(pk, cm, proof_slots=16, staged_load=True)
    t_load = time.perf_counter() - t0
    stop = threading.Event()
    made = []                 # (finished at, warm-up arrangement in force when it started, ok)
    lock = threading.Lock()

    def caller(i):
        k = i
        while not stop.is_set():
            warm = p.info()["warmup"]
            j = k % len(cases)
            ok = p.prove(w, *cases[j]).data == exp[j]
            with lock:
                made.append((time.perf_counter() - t0, warm, ok))
            k += 1
    ts = [threading.Thread(target=caller, args=(i,)) for i in range(4)]
    try:
        for t in ts:
            t.start()
        assert p.wait_ready(180_000)
        t_ready = time.perf_counter() - t0
        time.sleep(1.0)
        stop.set()
        for t in ts:
            t.join()
        lt, info = p.load_timings(), p.info()
        assert all(ok for _, _, ok in made), [x for x in made if not x[2]][:3]
        n_warm = sum(1 for _, warm, _ in made if warm)
        assert n_warm >= 1 and len(made) - n_warm >= 8, (n_warm, len(made))
        assert info["warmup"] == 0 and info["proof_slots"] == 16 and lt["background_status"] == 0
        first = min(t for t, _, _ in made)
        print("\n[staged S21] load returned after %.3f s, first proof done at %.3f s, ready at %.3f s; %d warm-up proofs, %d after; "
              "windows from a warm-up proof: %d, tuned: %d; timings %s" %
              (t_load, first, t_ready, n_warm, len(made) - n_warm, lt["windows_from_proof"], info["tuned"], json.dumps(lt)))
    finally:
        stop.set()
        for t in ts:
            t.join()
        p.close()
