# (the environment switches exist in the tuning build only: `python crescent-credentials_amd/build.py --tuning`)
# usage: tools/ab_env.sh VAR [reps]: the default headline measurement with VAR unset / VAR=1, alternating, on one box
set -u
V=$1; N=${2:-4}
for i in $(seq $N); do
  CRESCENT_GPU_LIB=crescent-credentials_amd/libcrescent_gpu_tuning.so python bench.py --witness device --steps 100 --headline-only --no-sweep --no-cpu-baseline --no-check 2>/dev/null | python tools/line_value.py "default"
  env $V=1 CRESCENT_GPU_LIB=crescent-credentials_amd/libcrescent_gpu_tuning.so python bench.py --witness device --steps 100 --headline-only --no-sweep --no-cpu-baseline --no-check 2>/dev/null | python tools/line_value.py "$V=1"
done
