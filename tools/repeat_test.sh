# usage: tools/repeat_test.sh N "<pytest node id>" [ENV=VAL ...]: run one test N times in fresh processes, report the exit codes
# (hunting an intermittent failure: which runs die, with what on stderr)
set -u
N=$1; T=$2; shift; shift
ok=0; bad=0
for i in $(seq $N); do
  env "$@" python -m pytest "$T" -x -q -p no:cacheprovider > /tmp/rt_$i.log 2>&1
  rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "run $i rc=$rc"; grep -v "^  File\|^$" /tmp/rt_$i.log | head -12; fi
done
echo "[$*] $T: $ok ok, $bad failed of $N"
