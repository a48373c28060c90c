set -u
O=gpurun_out/r05_ag; mkdir -p $O
export CRESCENT_GPU_LIB=$PWD/crescent-credentials_amd/libcrescent_gpu_tuning.so
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-host-witness"
run() { label="$1"; shift; env "$@" $B 2>>$O/err.log | python tools/line_value.py "$label"; }
(for i in 1 2 3; do
 run "two transform tiles per CU (shipped)" X=1
 run "one transform tile per CU (16 KB of unused LDS)" CG_NTT_LDS_PAD=16384
done) 2>&1 | tee $O/ntt_one_tile_per_cu.txt
grep -c "verifies: True" $O/err.log
