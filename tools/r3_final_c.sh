set -u
O=gpurun_out/r3final_c; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_and_ranks.py -x -q 2>&1 | tail -2)
tools/profile_pmc.sh $O/pmc "rs256-sd/gates/bits=0.90" > /dev/null 2>&1
cp $O/pmc/pmc_counters.json profiles/pmc_counters.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
tools/profile_serial.sh $O/serial > /dev/null 2>&1
for s in 1 8; do python tools/probe_latency.py $s 2>/dev/null | cut -c1-330; done > $O/latency.txt
ls $O $O/pmc
