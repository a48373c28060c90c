"""N > 1 path on CPU: two `gloo` ranks run the sharded-proof protocol of crescent_credentials_amd.distributed
(range-shard every query, all_gather 384-byte partial records, assemble) with the ORACLE standing in for the
GPU shard, and must reproduce the golden proof of the unsharded prover.  Also covers the rank-timing reduction
bench.py uses."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _rows(j):
    return tuple([[(int(c, 16), col) for c, col in row] for row in mat] for mat in j)


def _unpack_pk(o, j):
    g1s = lambda h: [o.g1_unpack(bytes.fromhex(h)[i:i + 64]) for i in range(0, len(h) // 2, 64)]
    g2s = lambda h: [o.g2_unpack(bytes.fromhex(h)[i:i + 128]) for i in range(0, len(h) // 2, 128)]
    vk = dict(alpha_g1=g1s(j["alpha_g1"])[0], beta_g2=g2s(j["beta_g2"])[0], gamma_g2=g2s(j["gamma_g2"])[0],
              delta_g1=g1s(j["delta_g1"])[0], delta_g2=g2s(j["delta_g2"])[0], gamma_abc_g1=g1s(j["gamma_abc_g1"]))
    return dict(vk=vk, beta_g1=g1s(j["beta_g1"])[0], delta_g1=vk["delta_g1"], a_query=g1s(j["a_query"]),
                b_g1_query=g1s(j["b_g1_query"]), b_g2_query=g2s(j["b_g2_query"]), h_query=g1s(j["h_query"]),
                l_query=g1s(j["l_query"]))


class OracleShard:
    """Stand-in for a GPU shard: same contract as Prover.prove_partial / Prover.assemble, computed by the oracle."""

    def __init__(self, o, pk, mats, l, m, M, rank, count, span=None):
        from crescent_credentials_amd.distributed import shard_range
        self.o, self.pk, self.mats, self.l, self.m, self.M = o, pk, mats, l, m, M
        self.count = count
        D = o.domain_size_for(m + l)
        # (the stand-in keeps the key as loaded: its "coset values" are the D - 1 coefficients of h, and a span cuts those)
        self.domain_size = D - 1
        part = (lambda n: (n * span[0] // 10000, n * span[1] // 10000)) if span else (lambda n: shard_range(n, rank, count))
        self.rh, self.rl, self.ra = part(D - 1), part(M - l), part(M - 1)

    def prove_partial(self, w, r, on_device=False, h_slice=None):
        o, pk, l = self.o, self.pk, self.l
        (h0, h1), (l0, l1), (a0, a1) = self.rh, self.rl, self.ra
        if h_slice is None:
            h_slice = o.witness_map_from_matrices(self.mats, l, self.m, w)[h0:h1]
            self.map_calls = getattr(self, "map_calls", 0) + 1
        assert len(h_slice) == h1 - h0
        ph = o.G1.to_affine(o.G1.msm(pk["h_query"][h0:h1], h_slice))
        pl = o.G1.to_affine(o.G1.msm(pk["l_query"][l0:l1], w[l + l0:l + l1]))
        pa = o.G1.to_affine(o.G1.msm(pk["a_query"][1 + a0:1 + a1], w[1 + a0:1 + a1]))
        pb1 = None if r == 0 else o.G1.to_affine(o.G1.msm(pk["b_g1_query"][1 + a0:1 + a1], w[1 + a0:1 + a1]))
        pb2 = o.G2.to_affine(o.G2.msm(pk["b_g2_query"][1 + a0:1 + a1], w[1 + a0:1 + a1]))
        return o.g1_packed(ph) + o.g1_packed(pl) + o.g1_packed(pa) + o.g1_packed(pb1) + o.g2_packed(pb2)

    # the "scatter" arrangement's three calls (cg_witness_map_coset / cg_h_scalars_slice / cg_prove_partial_q).  The stand-in
    # keeps the key as loaded, so "all the h scalars" are the coefficients of h and a shard's slice is its range of them
    def witness_map_coset(self, w, on_device=False, out_dev=None):
        h = self.o.witness_map_from_matrices(self.mats, self.l, self.m, w)
        self.coset_calls = getattr(self, "coset_calls", 0) + 1
        return b"".join(self.o.fe_bytes(x) for x in h)

    def h_scalars_slice(self, shard):
        from crescent_credentials_amd.distributed import shard_range
        D = self.o.domain_size_for(self.m + self.l)
        lo, hi = shard_range(D - 1, shard, self.count)
        return lo, hi - lo

    def prove_partial_q(self, w, q_slice, r, on_device=False, q_on_device=False):
        q = bytes(q_slice)
        hs = [int.from_bytes(q[i:i + 32], "little") for i in range(0, len(q), 32)]
        return self.prove_partial(w, r, h_slice=hs)

    # the two-call form (cg_prove_partial_q_begin / cg_partial_witness_map_coset / cg_prove_partial_q_finish)
    def prove_partial_q_begin(self, w, r, on_device=False):
        shard = self
        shard.begun = getattr(shard, "begun", 0) + 1

        class Open:
            def witness_map_coset(self, out_dev=None, out_host=None):
                return shard.witness_map_coset(w)

            def finish(self, q_slice, q_on_device=False):
                shard.finished = getattr(shard, "finished", 0) + 1
                return shard.prove_partial_q(w, q_slice, r)

            # the witness map in two halves: the stand-in's "a side" is its h vector, its "b side" all ones - what matters
            # here is that the two sides travel separately and are multiplied slice by slice
            def witness_map_coset_half(self, which, out_dev=None, out_host=None):
                shard.half_calls = getattr(shard, "half_calls", []) + [which]
                if which == 0:
                    return shard.witness_map_coset(w)
                D = shard.o.domain_size_for(shard.m + shard.l)
                return shard.o.fe_bytes(1) * (D - 1)

            def finish2(self, a_slice, b_slice, on_device=False):
                shard.finished = getattr(shard, "finished", 0) + 1
                a, b = bytes(a_slice), bytes(b_slice)
                assert len(a) == len(b)
                prod = b"".join(shard.o.fe_bytes(int.from_bytes(a[i:i + 32], "little") * int.from_bytes(b[i:i + 32], "little") % shard.o.R)
                                for i in range(0, len(a), 32))
                return shard.prove_partial_q(w, prod, r)

            def abort(self):
                shard.aborted = getattr(shard, "aborted", 0) + 1
        return Open()

    def assemble(self, parts, n, r, s):
        o, pk = self.o, self.pk
        acc = [o.G1.jac_infinity() for _ in range(4)]
        acc2 = o.G2.jac_infinity()
        for k in range(n):
            p = parts[384 * k:384 * (k + 1)]
            for i in range(4):
                acc[i] = o.G1.add_affine(acc[i], o.g1_unpack(p[64 * i:64 * i + 64]))
            acc2 = o.G2.add_affine(acc2, o.g2_unpack(p[256:384]))
        h_acc, l_acc, a_acc, b1_acc = acc
        d1 = o.G1.to_jac(pk["delta_g1"])
        r_g1 = o.G1.mul(d1, r)
        rs_delta = o.G1.mul(r_g1, s)
        g_a = o.G1.add_affine(o.G1.add(o.G1.add_affine(r_g1, pk["a_query"][0]), a_acc), pk["vk"]["alpha_g1"])
        s_g_a = o.G1.mul(g_a, s)
        if r != 0:
            g1_b = o.G1.add_affine(o.G1.add(o.G1.add_affine(o.G1.mul(d1, s), pk["b_g1_query"][0]), b1_acc), pk["beta_g1"])
        else:
            g1_b = o.G1.jac_infinity()
        g2_b = o.G2.add_affine(o.G2.add(o.G2.add_affine(o.G2.mul(o.G2.to_jac(pk["vk"]["delta_g2"]), s), pk["b_g2_query"][0]), acc2),
                               pk["vk"]["beta_g2"])
        g_c = o.G1.add(s_g_a, o.G1.mul(g1_b, r))
        g_c = o.G1.add(g_c, o.G1.neg(rs_delta))
        g_c = o.G1.add(o.G1.add(g_c, l_acc), h_acc)
        return o.proof_uncompressed((o.G1.to_affine(g_a), o.G2.to_affine(g2_b), o.G1.to_affine(g_c)))


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bn254_oracle as o
    from crescent_credentials_amd.distributed import ShardedProver, barrier_sync, max_over_ranks
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        g = load_golden("groth16_d8.json")
        pk = _unpack_pk(o, g["pk"])
        w = [int(x, 16) for x in g["witness"]]
        shard = OracleShard(o, pk, _rows(g["matrices"]), g["num_inputs"], g["num_constraints"], g["num_variables"], rank, world)
        sp = ShardedProver(shard, torch.device("cpu"))
        ok = True
        for case in g["proofs"]:
            proof = sp.prove(w, int(case["r"], 16), int(case["s"], 16))
            ok = ok and proof.hex() == case["proof"]
        ok = ok and sp.all_gathers == len(g["proofs"])          # exactly one collective per proof
        # several sharded proofs in flight per rank (K = 4): worker threads finish their partial sums in any order, the
        # records are exchanged strictly in proof order, every proof is the golden one
        cases = g["proofs"] * 3
        jobs = [(w, int(c["r"], 16), int(c["s"], 16)) for c in cases]
        times = []
        got = sp.prove_stream(jobs, in_flight=4, on_device=False, done_times=times)
        ok = ok and [p.hex() for p in got] == [c["proof"] for c in cases]
        ok = ok and sp.all_gathers == len(g["proofs"]) + len(jobs) and len(times) == len(jobs) and all(t > 0 for t in times)
        ok = ok and sp.prove_stream([], in_flight=4) == []
        # a shard that fails on ONE rank: no rank is left waiting in a collective, every collective is still issued in order,
        # the failing rank raises its own exception and the other names the rank and the job
        class Flaky:
            def __init__(self, inner, fail_at):
                self.inner, self.fail_at, self.calls = inner, fail_at, 0
            def prove_partial(self, a, r, on_device=False):
                self.calls += 1
                if (a, r) == self.fail_at:
                    raise ValueError("shard fault injected on rank %d" % rank)
                return self.inner.prove_partial(a, r, on_device=on_device)
            def assemble(self, *args):
                return self.inner.assemble(*args)
        jobs2 = [(w, 1000 + i, 7) for i in range(5)]
        spf = ShardedProver(Flaky(shard, (w, 1002) if rank == 1 else None), torch.device("cpu"))
        try:
            spf.prove_stream(jobs2, in_flight=3, on_device=False)
            ok = False
        except ValueError as e:
            ok = ok and rank == 1 and "injected" in str(e)
        except RuntimeError as e:
            ok = ok and rank != 1 and "rank 1 failed on job 2" in str(e)
        ok = ok and spf.all_gathers == len(jobs2)
        # the same for ONE proof at a time (ShardedProver.prove): the failing rank raises its own exception after the
        # collective, the others name it; nobody hangs, and the next proof goes through
        spo = ShardedProver(Flaky(shard, (w, 2002) if rank == 1 else None), torch.device("cpu"))
        try:
            spo.prove(w, 2002, 7)
            ok = False
        except ValueError as e:
            ok = ok and rank == 1 and "injected" in str(e)
        except RuntimeError as e:
            ok = ok and rank != 1 and "rank 1 failed" in str(e)
        c0_ = g["proofs"][0]
        ok = ok and spo.prove(w, int(c0_["r"], 16), int(c0_["s"], 16)).hex() == c0_["proof"] and spo.all_gathers == 2
        # SURVEY 8e's other arrangement: the witness map runs on rank 0 only and a scatter hands out the h scalars
        # (unequal slices here: the golden circuit's 7 h points over `world` ranks) - the same golden proofs, one
        # scatter + one all_gather per proof, and no rank but 0 ever runs the witness map
        shard_b = OracleShard(o, pk, _rows(g["matrices"]), g["num_inputs"], g["num_constraints"], g["num_variables"], rank, world)
        sps = ShardedProver(shard_b, torch.device("cpu"), arrangement="scatter")
        for case in g["proofs"]:
            ok = ok and sps.prove(w, int(case["r"], 16), int(case["s"], 16)).hex() == case["proof"]
        ok = ok and sps.all_gathers == sps.scatters == len(g["proofs"]) and getattr(shard_b, "map_calls", 0) == 0
        ok = ok and getattr(shard_b, "coset_calls", 0) == (len(g["proofs"]) if rank == 0 else 0)
        # the two-call form of the same arrangement: every rank opens the proof (its assignment-driven sums start), the source's
        # witness map runs on the open proof, the h share follows the scatter; the same golden proofs
        shard_t = OracleShard(o, pk, _rows(g["matrices"]), g["num_inputs"], g["num_constraints"], g["num_variables"], rank, world)
        spt = ShardedProver(shard_t, torch.device("cpu"), arrangement="scatter", two_call=True)
        for case in g["proofs"]:
            ok = ok and spt.prove(w, int(case["r"], 16), int(case["s"], 16)).hex() == case["proof"]
        ok = ok and spt.all_gathers == spt.scatters == len(g["proofs"]) == shard_t.begun == shard_t.finished
        ok = ok and getattr(shard_t, "aborted", 0) == 0 and getattr(shard_t, "coset_calls", 0) == (len(g["proofs"]) if rank == 0 else 0)
        # ... with the witness map in two halves: rank 0 computes one side, rank 1 the other, two scatters, the products on the shards
        shard_h = OracleShard(o, pk, _rows(g["matrices"]), g["num_inputs"], g["num_constraints"], g["num_variables"], rank, world)
        sph = ShardedProver(shard_h, torch.device("cpu"), arrangement="scatter", two_call=True, split_map=True)
        for case in g["proofs"]:
            ok = ok and sph.prove(w, int(case["r"], 16), int(case["s"], 16)).hex() == case["proof"]
        ok = ok and sph.all_gathers == len(g["proofs"]) and sph.scatters == 2 * len(g["proofs"])
        want_halves = ([0] if rank == 0 else []) + ([1] if rank == min(1, world - 1) else [])
        ok = ok and getattr(shard_h, "half_calls", []) == want_halves * len(g["proofs"]) and getattr(shard_h, "aborted", 0) == 0
        # ... and with UNEQUAL shares (cg_options.shard_span): the ranks that compute the halves carry less of the MSMs
        cuts = [0] + [10000 * (2 * k + 1) // (2 * world) for k in range(1, world)] + [10000]     # half a share for rank 0, the rest equal steps
        spans = [(cuts[k], cuts[k + 1]) for k in range(world)]
        shard_w = OracleShard(o, pk, _rows(g["matrices"]), g["num_inputs"], g["num_constraints"], g["num_variables"], rank, world, span=spans[rank])
        spw = ShardedProver(shard_w, torch.device("cpu"), arrangement="scatter", two_call=True, split_map=True, spans=spans)
        for case in g["proofs"]:
            ok = ok and spw.prove(w, int(case["r"], 16), int(case["s"], 16)).hex() == case["proof"]
        ok = ok and spw.all_gathers == len(g["proofs"]) and spw.scatters == 2 * len(g["proofs"])
        # a half that FAILS on its source (the b side, computed before either scatter: HScalarScatter.compute keeps the failure):
        # both scatters and the gather still happen everywhere, every rank gives its slot back, the failing rank raises its own
        # exception and the others name it; the next proof is the golden one
        class FlakyHalf:
            def __init__(self, inner, fail_side):
                self.inner, self.fail_side = inner, fail_side
            def prove_partial_q_begin(self, w_, r_, on_device=False):
                opened, fail_side = self.inner.prove_partial_q_begin(w_, r_, on_device=on_device), self.fail_side
                real = opened.witness_map_coset_half
                def half(which, **kw):
                    if which == fail_side:
                        raise ValueError("half %d fault injected on rank %d" % (which, rank))
                    return real(which, **kw)
                opened.witness_map_coset_half = half
                return opened
            def __getattr__(self, name):
                return getattr(self.inner, name)
        shard_f = OracleShard(o, pk, _rows(g["matrices"]), g["num_inputs"], g["num_constraints"], g["num_variables"], rank, world)
        flaky = FlakyHalf(shard_f, 1)
        spf2 = ShardedProver(flaky, torch.device("cpu"), arrangement="scatter", two_call=True, split_map=True)
        src_b = min(1, world - 1)
        try:
            spf2.prove(w, 3003, 7)
            ok = False
        except ValueError as e:
            ok = ok and rank == src_b and "half 1 fault injected" in str(e)
        except RuntimeError as e:
            ok = ok and rank != src_b and ("rank %d failed" % src_b) in str(e)
        ok = ok and spf2.scatters == 2 and spf2.all_gathers == 1
        ok = ok and getattr(shard_f, "aborted", 0) == (1 if rank == src_b else 0) and getattr(shard_f, "finished", 0) == (0 if rank == src_b else 1)
        flaky.fail_side = -1
        ok = ok and spf2.prove(w, int(c0_["r"], 16), int(c0_["s"], 16)).hex() == c0_["proof"]
        # ... and a STREAM of such proofs, the rank that runs the witness map rotating from job to job (k mod world): scatters
        # from one thread, gathers from another (two groups), three proofs in flight per rank; every proof the golden one, one
        # scatter + one gather per job, and the witness maps spread over the ranks
        shard_c = OracleShard(o, pk, _rows(g["matrices"]), g["num_inputs"], g["num_constraints"], g["num_variables"], rank, world)
        spr = ShardedProver(shard_c, torch.device("cpu"), arrangement="scatter", rotate=True, stream_slots=3)
        jobs_r = [(w, int(c["r"], 16), int(c["s"], 16)) for c in g["proofs"] * 4]
        times_r = []
        got_r = spr.prove_stream(jobs_r, in_flight=3, on_device=False, done_times=times_r)
        ok = ok and [p_.hex() for p_ in got_r] == [c["proof"] for c in g["proofs"] * 4]
        ok = ok and spr.scatters == spr.all_gathers == len(jobs_r) and all(t > 0 for t in times_r)
        mine_maps = len([k for k in range(len(jobs_r)) if k % world == rank])
        ok = ok and getattr(shard_c, "coset_calls", 0) == mine_maps and getattr(shard_c, "map_calls", 0) == 0
        # a witness map that fails on its source rank poisons that one job for everybody; the stream goes on
        class FlakyMap:
            def __init__(self, inner, fail_on_call):
                self.inner, self.fail_on_call, self.calls = inner, fail_on_call, 0
            def witness_map_coset(self, w_, on_device=False, out_dev=None):
                self.calls += 1
                if self.calls == self.fail_on_call:
                    raise ValueError("witness map fault injected on rank %d" % rank)
                return self.inner.witness_map_coset(w_, on_device=on_device)
            def __getattr__(self, name):
                return getattr(self.inner, name)
        spq = ShardedProver(FlakyMap(shard_c, 1 if rank == 1 else -1), torch.device("cpu"), arrangement="scatter", rotate=True, stream_slots=2)
        try:
            spq.prove_stream(jobs_r[:6], in_flight=2, on_device=False)
            ok = False
        except ValueError as e:
            ok = ok and rank == 1 and "injected" in str(e)
        except RuntimeError as e:
            ok = ok and rank != 1 and "rank 1 failed on job 1" in str(e)
        ok = ok and spq.scatters == spq.all_gathers == 6
        from crescent_credentials_amd.distributed import control_group, gather_over_ranks, min_over_ranks
        ok = ok and control_group() is dist.group.WORLD          # gloo default group IS the control plane
        # (collectives are evaluated on every rank whatever `ok` holds: a short-circuit would leave the others waiting)
        gathered_vals = gather_over_ranks(float(10 + rank), world)
        lowest = min_over_ranks(float(rank + 1), world)
        ok = ok and gathered_vals == [10.0 + k for k in range(world)] and lowest == 1.0
        # the data plane: RCCL cannot come up here (no GPU), which every rank finds out within the deadline, agrees on over
        # the control plane, and answers by handing back the gloo group with the reason; "gloo" asked for is gloo, no error
        from crescent_credentials_amd.distributed import open_data_group
        grp, used, err = open_data_group(torch.device("cuda", 0), "nccl", deadline_s=60.0)
        ok = ok and grp is control_group() and used == "gloo" and isinstance(err, str) and len(err) > 0
        grp2, used2, err2 = open_data_group(torch.device("cpu"), "gloo")
        ok = ok and grp2 is control_group() and used2 == "gloo" and err2 is None
        sp2 = ShardedProver(shard, torch.device("cpu"), group=grp)
        c0 = g["proofs"][0]
        again = sp2.prove(w, int(c0["r"], 16), int(c0["s"], 16)).hex()
        ok = ok and again == c0["proof"]
        barrier_sync(world)
        mx = max_over_ranks(float(rank + 1), world, torch.device("cpu"))
        q.put((rank, ok, mx))
    finally:
        dist.destroy_process_group()


def test_shard_ranges_partition():
    from crescent_credentials_amd.distributed import shard_range
    for n in (0, 1, 7, 1023, 1499999):
        for count in (1, 2, 3, 8):
            rs = [shard_range(n, k, count) for k in range(count)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[k][1] == rs[k + 1][0] for k in range(count - 1))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_proof_gloo_ranks(world):
    """world 8: the size of the node the multi-GPU configurations are quoted on (the golden circuit's h query has 7 points,
    so at eight ranks some shards are EMPTY - their partial sums are identities and the proof must not change)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in res) == list(range(world))
    assert all(ok for _, ok, _ in res), "sharded proof differs from the golden proof"
    assert all(mx == float(world) for _, _, mx in res)
