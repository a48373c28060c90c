"""Child process of test_a_retune_that_fails_half_way_refuses_every_later_proof_even_callers_already_waiting: runs against the
TUNING build of the library (CRESCENT_GPU_LIB), the only build that carries the CG_FAULT_RETUNE fault injector.

CG_FAULT_RETUNE=1 makes the re-size of the proof slots fail as an allocation would (table rebuilt, engines not), with several
callers in flight: every call returns either the right bytes (it finished before the re-tune) or CG_ERR_OUT_OF_MEMORY with
the reload message - never other bytes, never a fault - and once one call has been refused every later one is."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

SEED = 0xC5E5CE47


def main():
    from concurrent.futures import ThreadPoolExecutor
    import bn254_oracle as oracle
    import cpu_ref
    import crescent_credentials_amd as cc
    from crescent_credentials_amd import workloads as wl
    assert cc.library_path().endswith("libcrescent_gpu_tuning.so"), cc.library_path()
    assert cc.lib().cg_init(0, None) == 0
    l, m, M = wl.SHAPES["medium"]
    cm, w = wl.synthetic_circuit(SEED + 5, l, m, M, 0.9, 3, profile="gates")
    rng = random.Random(SEED + 5)
    pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, oracle.R) for _ in range(4)])
    want = cpu_ref.prove(pk, (cm.a, cm.b, cm.c), l, m, M, w, 5, 6, nthreads=8)
    prover = cc.Prover(pk, cm, proof_slots=4)
    os.environ["CG_FAULT_RETUNE"] = "1"
    try:
        def one(_):
            try:
                return prover.prove(w, 5, 6).data
            except cc.CrescentGpuError as e:
                return e
        with ThreadPoolExecutor(max_workers=6) as ex:
            got = list(ex.map(one, range(24)))
        good = [g for g in got if isinstance(g, bytes)]
        bad = [g for g in got if not isinstance(g, bytes)]
        assert good and all(g == want for g in good)                 # the proof that triggered the re-tune is itself fine
        assert bad and all(e.code == -4 and "load the circuit again" in str(e) for e in bad)
        first_bad = next(i for i, g in enumerate(got) if not isinstance(g, bytes))
        assert len(good) <= first_bad + 6                            # nothing is proved once the context is refused
        for _ in range(3):
            try:
                prover.prove(w, 5, 6)
                raise AssertionError("a refused context proved")
            except cc.CrescentGpuError:
                pass
        try:
            prover.witness_map(w)
            raise AssertionError("a refused context ran the witness map")
        except cc.CrescentGpuError:
            pass
    finally:
        del os.environ["CG_FAULT_RETUNE"]
        prover.close()
    # the same circuit loads and proves again
    p2 = cc.Prover(pk, cm, proof_slots=2)
    try:
        assert p2.prove(w, 5, 6).data == want and p2.prove(w, 5, 6).data == want and p2.info()["tuned"] == 1
    finally:
        p2.close()
    # CG_FAULT_STAGED=1: the WORKER of a staged load fails while it builds the final arrangement (as an allocation of the final
    # tables would beside another tenant of the GPU).  The context keeps proving in the warm-up arrangement - the right bytes,
    # from several callers - cg_ctx_wait_ready reports the failure with its message, and the context frees cleanly.
    os.environ["CG_FAULT_STAGED"] = "1"
    try:
        p3 = cc.Prover(pk, cm, proof_slots=4, staged_load=True)
        assert p3.prove(w, 5, 6).data == want
        try:
            p3.wait_ready(120_000)
            raise AssertionError("wait_ready did not report the worker's failure")
        except cc.CrescentGpuError as e:
            assert e.code == -4 and "warm-up arrangement" in str(e) and "injected" in str(e)
        lt, info = p3.load_timings(), p3.info()
        assert lt["staged"] == 1 and lt["ready"] == 0 and lt["background_status"] == -4 and info["warmup"] == 1
        with ThreadPoolExecutor(max_workers=4) as ex:
            assert all(x == want for x in ex.map(lambda _: p3.prove(w, 5, 6).data, range(12)))
        p3.close()
    finally:
        del os.environ["CG_FAULT_STAGED"]
    print("FAULT-RETUNE-OK")


if __name__ == "__main__":
    main()
