set -u
O=gpurun_out/r05_aj; mkdir -p $O
(for t in tests/test_gpu_host_and_ranks.py::test_contexts_come_and_go_while_others_prove tests/test_gpu_parity.py::test_concurrent_proofs_on_one_context_and_across_contexts tests/test_gpu_parity.py::test_proofs_in_flight_are_independent tests/test_gpu_parity.py::test_rejected_witnesses_among_proofs_in_flight; do
  bash tools/repeat_test.sh 8 $t
done) 2>&1 | tee $O/repeat.txt
