set -u
O=gpurun_out/r05_ah; mkdir -p $O
D=$PWD/crescent-credentials_amd
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-host-witness"
run() { label="$1"; shift; env "$@" $B 2>>$O/err.log | python tools/line_value.py "$label"; }
(for i in 1 2 3; do
 run "shipped" X=1
 run "transform passes at wave priority 3" CRESCENT_GPU_LIB=$D/libcrescent_gpu_prio_NTT.so
 run "combine / bucket reduction kernels at wave priority 3" CRESCENT_GPU_LIB=$D/libcrescent_gpu_prio_TAIL.so
 run "both" CRESCENT_GPU_LIB=$D/libcrescent_gpu_prio_BOTH.so
done) 2>&1 | tee $O/wave_priority.txt
grep -c "verifies: True" $O/err.log
