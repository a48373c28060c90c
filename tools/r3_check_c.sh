set -u
O=gpurun_out/r3c; mkdir -p $O
hw() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); h=d['host_witness']
print('$2', 'device', d['value'], 'pinned', h['pinned']['proofs_per_s'], 'pageable', h['pageable']['proofs_per_s'], 'clock', d['roofline_valu']['sustained_clock_ghz'])"; }
for i in 1 2; do
  python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline > $O/hw_sync_$i.json 2>/dev/null; hw $O/hw_sync_$i.json sync
  CG_UPLOAD_IN_STREAM=1 python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline > $O/hw_instream_$i.json 2>/dev/null; hw $O/hw_instream_$i.json in-stream
done
for s in 1 2 8; do python tools/probe_latency.py $s 2>/dev/null | cut -c1-400; done
GPU_MAX_HW_QUEUES=12 python tools/probe_latency.py 2 2>/dev/null | cut -c1-200
