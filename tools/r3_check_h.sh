set -u
export CG_BUILD_JOBS=16
trap 'python crescent-credentials_amd/build.py > /dev/null 2>&1' EXIT
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "msm or sharded or prove_golden or edge or share" 2>&1 | tail -3)
probe() { python tools/probe_msm.py --group 2 --k 20 --reps 4 $2 2>/dev/null | tail -1 | python -c "
import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('$1 $2', 'accum_g2_ms', round(d['accum_g2_ms'],3), 'msm_b2_ms', round(d['msm_b2_ms'],3), 'entries', d['entries_g2'])"; }
probe "pair 2 waves" ""
probe "pair 2 waves" "--bits 0.9"
for W in 3 4; do
  CG_HIPCC_EXTRA="-DCG_G2PAIR_WAVES=$W" python crescent-credentials_amd/build.py > /dev/null 2>&1
  probe "pair waves=$W" ""
  probe "pair waves=$W" "--bits 0.9"
done
