# usage: tools/ab_build.sh REPS "<extra hipcc flags>": the headline measurement with the default build and with the flagged build,
# alternating on one box (the library is rebuilt in place between runs; the default build is restored at the end)
set -u
N=$1; FLAGS=$2
export CG_BUILD_JOBS=16
# whatever ends this script, the default build is what is left in place
trap 'python crescent-credentials_amd/build.py > /dev/null 2>&1' EXIT
run() { python bench.py --witness device --steps 100 --headline-only --no-sweep --no-cpu-baseline --no-check 2>/dev/null | python tools/line_value.py "$1"; }
for i in $(seq $N); do
  python crescent-credentials_amd/build.py > /dev/null 2>&1; run default; run default
  CG_HIPCC_EXTRA="$FLAGS" python crescent-credentials_amd/build.py > /dev/null 2>&1; run "[$FLAGS]"; run "[$FLAGS]"
done
python crescent-credentials_amd/build.py > /dev/null 2>&1
