set -u
O=gpurun_out/r05_a; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err; echo
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python bench.py --witness device --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness > $O/bench_device.json 2> $O/bench_device.err
for f in $O/bench_default.json $O/bench_driver_args.json $O/bench_device.json; do python tools/line_value.py $f < $f; done
tools/profile_serial.sh $O/serial > /dev/null 2>&1; tail -3 $O/serial/efficiency.md
tools/profile_sq_pipelined.sh $O/sqpipe > $O/sqpipe.out 2>&1; tail -12 $O/sqpipe.out
(time python -m pytest tests -m gpu -q -x) > $O/gputests.log 2>&1; tail -5 $O/gputests.log
tools/profile_pcsamp.sh $O/pcsamp > $O/pcsamp.out 2>&1; tail -5 $O/pcsamp.out
