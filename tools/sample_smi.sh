#!/bin/bash
# Sample GPU clock / power while a command runs: tools/sample_smi.sh out.txt -- cmd...
out=$1; shift; shift
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|GPU use" | tr '\n' ' '; echo; sleep 0.05; done ) > "$out" &
SMI=$!
"$@"
kill $SMI 2>/dev/null
wait $SMI 2>/dev/null
