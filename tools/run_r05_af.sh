set -u
O=gpurun_out/r05_af; mkdir -p $O
(time python -m pytest tests -m gpu -q) > $O/gputests.log 2>&1; tail -3 $O/gputests.log
tools/profile_pmc.sh $O/pmc "rs256-sd/gates/bits=0.90" > /dev/null 2>&1
cp $O/pmc/pmc_counters.json profiles/pmc_counters.json && cp profiles/pmc_counters.json $O/pmc_counters.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err; echo
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
for f in bench_default bench_driver_args; do python tools/line_value.py $f < $O/$f.json; done
python -c "import json; d=json.load(open('$O/bench_default.json')); print(d['roofline_valu']['counters'], d['roofline']['traffic'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/probe_latency.py > $O/latency.txt 2>&1; tail -12 $O/latency.txt
python tools/soak.py 6000 16 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
