set -u
for s in 1 2 8; do python tools/probe_latency.py $s 2>/dev/null | cut -c1-330; done
for s in 1 8; do CG_G2_PAIR=0 python tools/probe_latency.py $s 2>/dev/null | cut -c1-120; done
