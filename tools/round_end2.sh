# the default bench line, the 2-rank gloo plumbing run, and the lone-proof / shard latencies on one box
set -u
O=gpurun_out/final2; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 24 --warmup 4 --backend gloo > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; tail -c 300 $O/bench_2rank_gloo.json
for s in 1 2 4 8; do python tools/probe_latency.py $s 2>/dev/null | cut -c1-200; done | tee $O/latency.txt
tools/profile_pipelined.sh $O/pipe > /dev/null 2>&1
