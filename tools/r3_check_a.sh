set -u
O=gpurun_out/r3a; mkdir -p $O
(time timeout 1700 python -m pytest tests/test_gpu_host_and_ranks.py tests/test_gpu_e2e_files.py -x -q) > $O/tests.log 2>&1; tail -15 $O/tests.log
(time timeout 900 python bench.py --steps 20 --warmup 5 --no-sweep) > $O/bench_20.json 2> $O/bench_20.err; tail -5 $O/bench_20.err; cut -c1-1500 $O/bench_20.json
