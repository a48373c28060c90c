# (the environment switches exist in the tuning build only)
# usage: tools/ab_envval.sh VAR "v1 v2 ..." [reps]: the default headline measurement with VAR unset and with VAR=each value, alternating, on one box
set -u
V=$1; VALS=$2; N=${3:-3}
for i in $(seq $N); do
  CRESCENT_GPU_LIB=$PWD/crescent-credentials_amd/libcrescent_gpu_tuning.so python bench.py --witness device --steps 100 --headline-only 2>/dev/null | python tools/line_value.py "default"
  for x in $VALS; do
    env $V=$x CRESCENT_GPU_LIB=$PWD/crescent-credentials_amd/libcrescent_gpu_tuning.so python bench.py --witness device --steps 100 --headline-only 2>/dev/null | python tools/line_value.py "$V=$x"
  done
done
