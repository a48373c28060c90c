set -u
O=gpurun_out/r05_g; mkdir -p $O
(time python -m pytest tests -m gpu -q) > $O/gputests.log 2>&1; tail -3 $O/gputests.log
tools/profile_pmc.sh $O/pmc "rs256-sd/gates/bits=0.90" > /dev/null 2>&1
cp $O/pmc/pmc_counters.json profiles/pmc_counters.json && cp profiles/pmc_counters.json $O/pmc_counters.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err; echo
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
for f in bench_default bench_driver_args; do python tools/line_value.py $f < $O/$f.json; done
python bench.py --gpus 2 --backend gloo --steps 24 --warmup 4 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; python tools/line_value.py 2rank < $O/bench_2rank_gloo.json
python bench.py --gpus 8 --backend gloo --allow-shared-gpu --inflight 1 --steps 8 --warmup 2 --blocks 3 --no-host-witness --no-check --sharded-steps 6 --sharded-inflight 2 --sharded-stream 16 --leg-timeout 1500 > $O/bench_8rank_gloo_S21.json 2> $O/bench_8rank_gloo_S21.err; python tools/line_value.py 8rank < $O/bench_8rank_gloo_S21.json; tail -c 400 $O/bench_8rank_gloo_S21.err
(for t in "tests/test_gpu_host_and_ranks.py::test_contexts_come_and_go_while_others_prove" "tests/test_gpu_parity.py::test_concurrent_proofs_on_one_context_and_across_contexts" "tests/test_gpu_parity.py::test_proofs_in_flight_are_independent" "tests/test_gpu_parity.py::test_rejected_witnesses_among_proofs_in_flight"; do tools/repeat_test.sh 12 "$t"; done) > $O/repeat_concurrency_tests.txt 2>&1; tail -4 $O/repeat_concurrency_tests.txt
python tools/soak.py 6000 12 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
