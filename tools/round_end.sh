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
