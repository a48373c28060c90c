set -u
O=gpurun_out/r05_aa; mkdir -p $O
python tools/soak.py 20000 16 > $O/soak.txt 2>&1; tail -3 $O/soak.txt
python bench.py --gpus 2 --backend gloo --steps 24 --warmup 4 --no-host-witness --no-check > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; python tools/line_value.py 2rank < $O/bench_2rank_gloo.json
python bench.py --gpus 8 --backend gloo --allow-shared-gpu --inflight 1 --steps 8 --warmup 2 --blocks 3 --no-host-witness --no-check --sharded-steps 6 --sharded-inflight 2 --sharded-stream 16 --leg-timeout 1500 > $O/bench_8rank_gloo_S21.json 2> $O/bench_8rank_gloo_S21.err; python tools/line_value.py 8rank < $O/bench_8rank_gloo_S21.json
