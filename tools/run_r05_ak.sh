set -u
O=gpurun_out/r05_ak; mkdir -p $O
D=$PWD/crescent-credentials_amd
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-host-witness"
run() { label="$1"; shift; env "$@" $B 2>>$O/err.log | python tools/line_value.py "$label"; }
(for i in 1 2 3; do
 run "shipped (default machine scheduler)" X=1
 run "-mllvm -amdgpu-sched-strategy=iterative-ilp" CRESCENT_GPU_LIB=$D/libcrescent_gpu_sched.so
done) 2>&1 | tee $O/sched_strategy.txt
grep -c "verifies: True" $O/err.log
