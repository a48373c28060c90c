set -u
mkdir -p gpurun_out/str
(time python -m pytest tests -m gpu -x -q -k "shard") > gpurun_out/str/tests.log 2>&1; tail -4 gpurun_out/str/tests.log
for s in 2 4 8; do
  python tools/probe_latency.py $s 2>/dev/null | sed "s/^/strided /"
  CG_CONTIGUOUS_SHARDS=1 python tools/probe_latency.py $s 2>/dev/null | sed "s/^/contiguous /"
done | tee gpurun_out/str/latency.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 24 --warmup 4 --backend gloo > gpurun_out/str/bench_2rank_gloo.json 2> gpurun_out/str/bench_2rank_gloo.err
tail -c 1200 gpurun_out/str/bench_2rank_gloo.json; tail -3 gpurun_out/str/bench_2rank_gloo.err
