# end-of-round artefacts on one box: full GPU test suite, default bench line, serial efficiency + pipelined kernel stats + PMC passes
set -u
O=gpurun_out/final
mkdir -p $O
(time python -m pytest tests -m gpu -q) > $O/gputests.log 2>&1; tail -3 $O/gputests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
tools/profile_serial.sh $O/serial > /dev/null 2>&1
tools/profile_pipelined.sh $O/pipe > /dev/null 2>&1
tools/profile_pmc.sh $O/pmc "rs256-sd/gates/bits=0.90" > /dev/null 2>&1
ls -la $O $O/serial $O/pipe $O/pmc
# the 2-rank plumbing run (gloo, both ranks on this one GPU) and the lone-proof / shard latencies
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 24 --warmup 4 --inflight 4 --backend gloo > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
for s in 1 2 4 8; do python tools/probe_latency.py $s 2>/dev/null | cut -c1-220; done > $O/latency.txt
