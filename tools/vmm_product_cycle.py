#!/usr/bin/env python3
"""GPU box: does the product give physical memory back?  Create and close a large solver several times in ONE process and print what the device reports
free each time (round 6: a destroyed solver's mapped arrays kept their physical memory as long as their address ranges stayed reserved)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latticeurbanwind_amd as luw
luw.load()
os.environ.setdefault("LUW_TUNE_PLACEMENT", "0")
size = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else [1024, 1024, 256]
print("device free at start %.1f GB" % (torch.cuda.mem_get_info(0)[0] / 1e9), flush=True)
for k in range(int(os.environ.get("CYCLES", "6"))):
    g = luw.LBM(*size, 1e-4)
    g.run(0); g.run(2)
    during = torch.cuda.mem_get_info(0)[0]
    g.close()
    print("cycle %d: free with the solver alive %.1f GB, after close %.1f GB" % (k, during / 1e9, torch.cuda.mem_get_info(0)[0] / 1e9), flush=True)
# the eight-domain tile of BASELINE configs[4] on this one device (about fifty mapped ranges, 60 GB), created and destroyed: what round 5's fault followed
for k in range(int(os.environ.get("GROUP_CYCLES", "6"))):
    g = luw.LBMGroup(2048, 1024, 512, 4, 2, 1, 1e-4, fp16c=True, devices=[0] * 8)
    g.run(0); g.run(2)
    during = torch.cuda.mem_get_info(0)[0]
    g.close()
    print("group cycle %d: free with the group alive %.1f GB, after close %.1f GB" % (k, during / 1e9, torch.cuda.mem_get_info(0)[0] / 1e9), flush=True)
