set -u
O=gpurun_out/r05_j; mkdir -p $O
(time python -m pytest tests/test_gpu_host_and_ranks.py -m gpu -q -x) > $O/ranks.log 2>&1; tail -3 $O/ranks.log
(time python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "sharded_contexts") > $O/fullsize_sharded.log 2>&1; tail -3 $O/fullsize_sharded.log
python bench.py --gpus 2 --backend gloo --steps 24 --warmup 4 --no-host-witness --no-check > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; python tools/line_value.py 2rank < $O/bench_2rank_gloo.json
python bench.py --gpus 8 --backend gloo --allow-shared-gpu --inflight 1 --steps 8 --warmup 2 --blocks 3 --no-host-witness --no-check --sharded-steps 6 --sharded-inflight 2 --sharded-stream 16 --leg-timeout 1500 > $O/bench_8rank_gloo_S21.json 2> $O/bench_8rank_gloo_S21.err; python tools/line_value.py 8rank < $O/bench_8rank_gloo_S21.json
python - <<PY
import json
for f in ("bench_2rank_gloo", "bench_8rank_gloo_S21"):
    d = json.load(open("$O/" + f + ".json")); sh = d["sharded"]
    print(f, d["value"], "one at a time:", {k: v.get("ms_per_proof") for k, v in sh.get("arrangements", {}).items() if isinstance(v, dict)},
          "in flight recompute", sh["in_flight"]["proofs_per_s"], "scatter rotating", sh["in_flight"].get("scatter_rotating"))
PY
python bench.py --shape mdl1 --no-sweep > $O/bench_mdl1.json 2> $O/bench_mdl1.err; python tools/line_value.py mdl1 < $O/bench_mdl1.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/line_value.py default16 < $O/bench_default.json
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; python tools/line_value.py driver_args < $O/bench_driver_args.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
