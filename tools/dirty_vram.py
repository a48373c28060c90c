"""Fill (nearly) all free device memory with a byte pattern and release it: what a fresh allocation then holds is that pattern
unless the driver clears it - an uninitialised read in a kernel shows up as a wrong result instead of passing on zeroed pages.
usage: python tools/dirty_vram.py [byte=0xA5] [leave_GB=6]"""
import sys
import torch
byte = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0xA5
leave = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
free, total = torch.cuda.mem_get_info(0)
bufs = []
chunk = 4 << 30
filled = 0
while True:
    free, _ = torch.cuda.mem_get_info(0)
    if free < leave * 1e9 + chunk:
        break
    bufs.append(torch.full((chunk,), byte, dtype=torch.uint8, device="cuda:0"))
    filled += chunk
torch.cuda.synchronize()
print("filled %.1f GB of %.1f GB with 0x%02X" % (filled / 1e9, total / 1e9, byte))
