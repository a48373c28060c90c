set -u
mkdir -p gpurun_out/ab1
export CG_BUILD_JOBS=16
run() {
  tag=$1
  python tools/probe_msm.py --k 20 --group 2 --bits 0.0 > gpurun_out/ab1/$tag.g2_u.txt 2>&1
  python tools/probe_msm.py --k 20 --group 2 --bits 0.9 > gpurun_out/ab1/$tag.g2_c.txt 2>&1
  python bench.py --steps 60 --no-sweep --no-cpu-baseline > gpurun_out/ab1/$tag.bench09.json 2> gpurun_out/ab1/$tag.bench09.err
  python bench.py --steps 40 --bits 0 --no-sweep --no-cpu-baseline > gpurun_out/ab1/$tag.bench00.json 2> gpurun_out/ab1/$tag.bench00.err
}
run two
CG_HIPCC_EXTRA="-DCG_MUL2_ONE_CHAIN" python crescent-credentials_amd/build.py > gpurun_out/ab1/build_one.log 2>&1
run one
python crescent-credentials_amd/build.py > gpurun_out/ab1/build_two.log 2>&1
run two_b
for f in gpurun_out/ab1/*.txt; do echo $f; tail -1 $f; done
for f in gpurun_out/ab1/*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['phase_ms'])"; done
