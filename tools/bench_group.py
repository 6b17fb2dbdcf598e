#!/usr/bin/env python3
"""Step time of the one-process multi-domain host (luw_group_*) with all domains on ONE GPU: (a) tiny domains -> what the single
enqueueing host thread costs per domain and step; (b) the 8-GPU tile's domains one after the other on one device -> kernel + shell +
pack / unpack work per domain without any wire (the domains serialise on the device, so ms/step/domain is what one GPU of a node does).
usage: bench_group.py [f32|fp16c] [quick]      (quick: the eight-domain cases only; LUW_GROUP_EXCHANGE=sequential for the three-phase exchange)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import latticeurbanwind_amd as luw
from bench import fill_channel, NU
luw.load()
fp16c = len(sys.argv) > 1 and sys.argv[1] == "fp16c"
quick = "quick" in sys.argv[2:]


def run(gN, D, steps, label):
    n = D[0] * D[1] * D[2]
    g = luw.LBMGroup(*gN, *D, NU, fp16c=fp16c, devices=[0] * n)
    fill_channel(g.flags, g.u, g.rho, *gN)
    g.run(0); g.run(5)
    t0 = time.perf_counter(); g.run(steps); dt = (time.perf_counter() - t0) / steps
    print("%-64s %8.3f ms/step  = %7.3f ms per domain   (%.0f MLUPS on this one GPU; overlap %s, peer stores %s)" % (label, dt * 1e3, dt * 1e3 / n,
        gN[0] * gN[1] * gN[2] / dt / 1e6, g.overlaps(), g.direct_peer_stores()) + " one-phase exchange %s" % g.one_phase())
    g.close()


if not quick:
    run((64, 32, 32), (1, 1, 1), 2000, "one tiny domain 64x32x32 (launch-bound)")
    run((128, 64, 32), (2, 2, 1), 1000, "four tiny domains [2,2,1] (host enqueue cost)")
run((256, 64, 64), (4, 2, 1), 1000, "eight tiny domains [4,2,1] (host enqueue cost)")
run((256, 64, 64), (1, 4, 2), 1000, "eight tiny domains [1,4,2]")
if not quick:
    run((512, 512, 512), (1, 1, 1), 60, "512^3 undivided")
    run((1024, 512, 512), (1, 2, 1), 40, "two domains [1,2,1] of 1024x512x512")
    run((1024, 1024, 512), (1, 2, 2), 30, "four domains [1,2,2] of 1024x1024x512")
run((2048, 1024, 512), (1, 4, 2), 20, "eight domains [1,4,2] of 2048x1024x512 (BASELINE configs[3])")
run((2048, 1024, 512), (4, 2, 1), 20, "eight domains [4,2,1] (the deck's literal grid)")
