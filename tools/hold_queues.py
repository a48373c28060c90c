"""Keeps N hardware queues of this process alive and idle (one HIP stream each, used once) for S seconds: the state the pytest
parent process is in - after ten minutes of GPU tests - when it starts the eight-process bench run as a child.
usage: GPU_MAX_HW_QUEUES=20 python tools/hold_queues.py [N=20] [S=600] [busy=0]"""
import sys
import time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
busy = int(sys.argv[3]) if len(sys.argv) > 3 else 0
streams = [torch.cuda.Stream(device=0) for _ in range(n)]
x = torch.zeros(1 << 20, device="cuda:0")
for s in streams:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
print("holding %d streams" % n, flush=True)
t0 = time.time()
while time.time() - t0 < secs:
    if busy:
        for s in streams:
            with torch.cuda.stream(s):
                x.add_(1.0)
        torch.cuda.synchronize()
    else:
        time.sleep(1.0)
