# end-of-round artefacts on one box: full GPU test suite, default bench line, serial efficiency + per-launch accumulation table +
# pipelined kernel stats + SQ counter passes + the PMC passes behind profiles/pmc_counters.json, the 2-rank run as the driver
# invokes it (gloo, both ranks on this one GPU), lone-proof / shard latencies.   usage: tools/round_end.sh [out-dir]
set -u
O=${1:-gpurun_out/final}
mkdir -p $O
(time python -m pytest tests -m gpu -q) > $O/gputests.log 2>&1; tail -3 $O/gputests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
tools/profile_serial.sh $O/serial > /dev/null 2>&1
tools/profile_pipelined.sh $O/pipe > /dev/null 2>&1
tools/profile_pmc.sh $O/pmc "rs256-sd/gates/bits=0.90" > /dev/null 2>&1
tools/profile_sq.sh $O/sq > /dev/null 2>&1
python bench.py --gpus 2 --backend gloo --steps 24 --warmup 4 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
for s in 1 2 4 8; do python tools/probe_latency.py $s 2>/dev/null | cut -c1-330; done > $O/latency.txt
ls -la $O $O/serial $O/pipe $O/pmc $O/sq
