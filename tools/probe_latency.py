"""wall-clock of one cg_prove_dev call at a time vs the library's own total_ms (host overhead outside the timed region)"""
import os, sys, time, random
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import crescent_credentials_amd as cc
from crescent_credentials_amd import workloads as wl
cc.lib().cg_init(0, None)
R = cc.api.FR_MODULUS
l, m, M = wl.SHAPES["rs256-sd"]
cm, w = wl.synthetic_circuit(3, l, m, M, 0.9, 3, profile="gates")
rng = random.Random(1)
pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, R) for _ in range(4)])
shard = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = cc.Prover(pk, cm, shard_rank=0, shard_count=shard) if shard > 1 else cc.Prover(pk, cm)
wd = torch.from_numpy(w).cuda()
f = (lambda t=False: p.prove_partial(wd.data_ptr(), 5, on_device=True, timings=t)) if shard > 1 else (lambda t=False: p.prove_dev(wd.data_ptr(), 5, 7, timings=t))
for _ in range(4):
    f()
ws, ts = [], []
for _ in range(10):
    t0 = time.perf_counter(); _, tm = f(True); ws.append((time.perf_counter() - t0) * 1e3); ts.append(tm["total_ms"])
if shard > 1:
    # SURVEY 8e's other arrangement, its two pieces alone on the GPU: the one witness map for all shards (cg_witness_map_coset,
    # device in / device out) and a shard that proves with its slice (cg_prove_partial_q on a context without witness-map memory)
    q = torch.empty(p.domain_size * 32, dtype=torch.uint8, device="cuda")
    ext = cc.Prover(pk, cm, shard_rank=1, shard_count=shard, h_scalars_external=True)
    off, cnt = p.h_scalars_slice(1)
    for _ in range(3):
        p.witness_map_coset(wd.data_ptr(), on_device=True, out_dev=q.data_ptr())
        ext.prove_partial_q(wd.data_ptr(), q.data_ptr() + off * 32, 5, on_device=True, q_on_device=True)
    tw, tq = [], []
    for _ in range(10):
        t0 = time.perf_counter(); p.witness_map_coset(wd.data_ptr(), on_device=True, out_dev=q.data_ptr()); tw.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); ext.prove_partial_q(wd.data_ptr(), q.data_ptr() + off * 32, 5, on_device=True, q_on_device=True); tq.append((time.perf_counter() - t0) * 1e3)
    print("shards", shard, "scatter arrangement: witness map for all shards", round(float(np.median(tw)), 3), "ms; a shard with its slice", round(float(np.median(tq)), 3), "ms")
    # the two-call form: on the witness-map rank the l, a, b sums are queued first and the witness map runs next to them; on every
    # rank only the h share is left once the slice has arrived (cg_prove_partial_q_begin / cg_partial_witness_map_coset / _finish)
    tsrc, tfin = [], []
    for i in range(13):
        t0 = time.perf_counter(); op = p.prove_partial_q_begin(wd.data_ptr(), 5, on_device=True); op.witness_map_coset(out_dev=q.data_ptr())
        t_src = (time.perf_counter() - t0) * 1e3
        o0, c0 = p.h_scalars_slice(0)
        op.finish(q.data_ptr() + o0 * 32, q_on_device=True)
        ope = ext.prove_partial_q_begin(wd.data_ptr(), 5, on_device=True)
        torch.cuda.synchronize()                       # its l, a, b1, b2 sums are done: what is left is what follows the scatter
        t0 = time.perf_counter(); ope.finish(q.data_ptr() + off * 32, q_on_device=True); t_fin = (time.perf_counter() - t0) * 1e3
        if i >= 3:
            tsrc.append(t_src); tfin.append(t_fin)
    print("shards", shard, "two-call form: begin + witness map on the source rank", round(float(np.median(tsrc)), 3), "ms; the h share after the slice",
          round(float(np.median(tfin)), 3), "ms; critical path ~", round(float(np.median(tsrc)) + float(np.median(tfin)), 3), "ms + the scatter")
    # the witness map in two halves: rank 0 computes the a side, rank 1 the b side (at the same time), two scatters, every shard
    # multiplies its slices and adds the h share (cg_partial_witness_map_coset_half / cg_prove_partial_q_finish2)
    qa = torch.empty(p.domain_size * 32, dtype=torch.uint8, device="cuda")
    qb = torch.empty(p.domain_size * 32, dtype=torch.uint8, device="cuda")
    th0, th1, tf2, talone = [], [], [], []
    for i in range(13):
        t0 = time.perf_counter(); op = p.prove_partial_q_begin(wd.data_ptr(), 5, on_device=True); op.witness_map_coset_half(0, out_dev=qa.data_ptr())
        t_h0 = (time.perf_counter() - t0) * 1e3
        op.witness_map_coset_half(1, out_dev=qb.data_ptr())
        o0, c0 = p.h_scalars_slice(0)
        op.finish2(qa.data_ptr() + o0 * 32, qb.data_ptr() + o0 * 32, on_device=True)
        t0 = time.perf_counter(); op = p.prove_partial_q_begin(wd.data_ptr(), 5, on_device=True); op.witness_map_coset_half(1, out_dev=qb.data_ptr())
        t_h1 = (time.perf_counter() - t0) * 1e3
        op.finish2(qa.data_ptr() + o0 * 32, qb.data_ptr() + o0 * 32, on_device=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); p.witness_map_coset_half(wd.data_ptr(), 0, on_device=True, out_dev=qa.data_ptr()); t_al = (time.perf_counter() - t0) * 1e3
        ope = ext.prove_partial_q_begin(wd.data_ptr(), 5, on_device=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ope.finish2(qa.data_ptr() + off * 32, qb.data_ptr() + off * 32, on_device=True); t_f2 = (time.perf_counter() - t0) * 1e3
        if i >= 3:
            th0.append(t_h0); th1.append(t_h1); tf2.append(t_f2); talone.append(t_al)
    md = lambda x: round(float(np.median(x)), 3)
    print("shards", shard, "two halves: begin + the a side on rank 0", md(th0), "ms, begin + the b side on rank 1", md(th1), "ms (one side alone on an idle GPU",
          md(talone), "ms); product + h share after the slices", md(tf2), "ms; critical path ~", round(max(md(th0), md(th1)) + md(tf2), 3), "ms + two scatters")
    # ... with UNEQUAL shares (cg_options.shard_span): the two ranks that compute a half carry `ws` of an equal share of the MSMs,
    # the others the rest; the pieces for ws = 0.6
    if shard > 2:
        ws = 0.6
        w_src = 10000.0 / shard * ws
        w_oth = (10000.0 - 2 * w_src) / (shard - 2)
        src_ctx = cc.Prover(pk, cm, shard_rank=0, shard_count=shard, shard_span=(0, int(w_src)))
        oth_ctx = cc.Prover(pk, cm, shard_rank=2, shard_count=shard, shard_span=(int(2 * w_src), int(2 * w_src + w_oth)), h_scalars_external=True)
        so, sc_ = src_ctx.h_scalars_slice(0)
        oo, oc = oth_ctx.h_scalars_slice(2)
        tsw, tow = [], []
        for i in range(13):
            t0 = time.perf_counter(); op = src_ctx.prove_partial_q_begin(wd.data_ptr(), 5, on_device=True); op.witness_map_coset_half(0, out_dev=qa.data_ptr())
            t_s = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter(); op.finish2(qa.data_ptr() + so * 32, qb.data_ptr() + so * 32, on_device=True); t_s2 = (time.perf_counter() - t0) * 1e3
            ope = oth_ctx.prove_partial_q_begin(wd.data_ptr(), 5, on_device=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); ope.finish2(qa.data_ptr() + oo * 32, qb.data_ptr() + oo * 32, on_device=True); t_o = (time.perf_counter() - t0) * 1e3
            if i >= 3:
                tsw.append((t_s, t_s2)); tow.append(t_o)
        s1, s2 = md([x[0] for x in tsw]), md([x[1] for x in tsw])
        print("shards", shard, "two halves, unequal shares (sources %.2f of an equal share): begin + a half on a source" % ws, s1, "ms, its product + h share", s2,
              "ms; another rank's product + h share", md(tow), "ms; critical path ~", round(s1 + max(s2, md(tow)), 3), "ms + two scatters")
        src_ctx.close(); oth_ctx.close()
    ext.close()
print("shards", shard, "wall ms", round(float(np.median(ws)), 3), "library total_ms", round(float(np.median(ts)), 3), {k: round(v, 3) for k, v in tm.items() if k.endswith("_ms")})
