set -u
O=gpurun_out/r05_ae; mkdir -p $O
export CRESCENT_GPU_LIB=$PWD/crescent-credentials_amd/libcrescent_gpu_tuning.so
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-host-witness"
run() { label="$1"; shift; env "$@" $B 2>>$O/err.log | python tools/line_value.py "$label"; }
(for i in 1 2 3; do
 run "accumulation in 256-thread workgroups (shipped)" X=1
 run "accumulation in 512-thread workgroups" CG_ACCUM_BLOCK=512
 run "accumulation in 1024-thread workgroups" CG_ACCUM_BLOCK=1024
done) 2>&1 | tee $O/accum_block_large.txt
grep -c "verifies: True" $O/err.log
