# eight tools/stress_load.py at once (shards 0..7 of one circuit) with two processes holding 24 idle hardware queues each
# usage: tools/stress_load.sh <loads per process> [env assignments for the loaders...]
set -u
# HOLDERS=0 in the environment: no holder processes (the control)
N=${1:-100}; shift
H1=""; H2=""
if [ "${HOLDERS:-2}" != "0" ]; then
  GPU_MAX_HW_QUEUES=24 python tools/hold_queues.py 24 3000 0 > /dev/null 2>&1 & H1=$!
  GPU_MAX_HW_QUEUES=24 python tools/hold_queues.py 24 3000 0 > /dev/null 2>&1 & H2=$!
  sleep 20
fi
PIDS=""
for r in 0 1 2 3 4 5 6 7; do
  env "$@" python tools/stress_load.py $r $N 2> /tmp/stress_load_$r.err & PIDS="$PIDS $!"
done
wait $PIDS
if [ -n "$H1" ]; then kill $H1 $H2; wait $H1 $H2 2>/dev/null; fi
# CG_CHECK_FOLD lines (tuning build): per process, the lines that differ from that process's most common one
python - <<'PY'
import collections, glob
for f in sorted(glob.glob("/tmp/stress_load_*.err")):
    lines = [x.strip() for x in open(f, errors="replace") if x.startswith("CGFOLD")]
    if not lines:
        continue
    common = collections.Counter(lines).most_common(1)[0][0].split()
    for k, ln in enumerate(lines):
        t = ln.split()
        if t != common:
            print(f, "load", k, "differs in:", " ".join(t[i - 1] for i in range(2, len(t), 2) if t[i] != common[i]))
    print(f, len(lines), "fold lines")
PY
