set -u
O=gpurun_out/r3b; mkdir -p $O
(time timeout 900 python bench.py) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
(time timeout 900 python bench.py --steps 20 --warmup 5 --no-sweep --no-cpu-baseline) > $O/bench_20.json 2> $O/bench_20.err; tail -2 $O/bench_20.err
(time timeout 900 python bench.py --gpus 2 --backend gloo --steps 24 --no-host-witness) > $O/bench_2rank.json 2> $O/bench_2rank.err; tail -3 $O/bench_2rank.err
python - <<'PY'
import json
for f in ("bench_default","bench_20","bench_2rank"):
    try:
        d=json.load(open("gpurun_out/r3b/%s.json"%f))
        print(f, d["value"], d["timing"]["spread_pct"], d["timing"]["bracketed"]["value"], d.get("host_witness",{}).get("pinned",{}).get("proofs_per_s"), d.get("host_witness",{}).get("pageable",{}).get("proofs_per_s"), json.dumps(d.get("sharded")))
    except Exception as e: print(f, "ERR", e)
PY
