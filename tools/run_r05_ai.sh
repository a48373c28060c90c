set -u
O=gpurun_out/r05_ai; mkdir -p $O
python bench.py --shape mdl1 --no-sweep > $O/bench_mdl1.json 2> $O/bench_mdl1.err; python tools/line_value.py mdl1 < $O/bench_mdl1.json; tail -2 $O/bench_mdl1.err
