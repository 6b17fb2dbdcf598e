#!/usr/bin/env python3
"""1024^3 on one MI355X (FP32: 82 GB of DDFs; FP16C: 41 GB, pair kernel): size-independent properties that need no oracle -- the rest
state is an exact fixed point, and a periodic box with a three-dimensional wave field conserves mass and stays finite.  Prints a digest
(xor and sum over the bit patterns of rho and u) so that two runs -- e.g. the flat and the row addressing form of the FP32 kernel
(LUW_TEST_AIDS=addr_row) -- can be held to identical results value for value.
usage: check_huge.py [f32|fp16c] [wave]     (wave: only the shear-wave part and its digest -- the second run of a pair)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import latticeurbanwind_amd as luw
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
N = 1024
t0 = time.time()
rest = True
if "wave" not in sys.argv[2:]:
    g = luw.LBM(N, N, N, 1e-5, fp16c=(dt == "fp16c"))
    g.run(4)
    g.rho.read_from_device(); g.u.read_from_device()
    rest = bool(np.all(g.rho.data == 1.0)) and not g.u.data.any()
    print("%s 1024^3 rest state after 4 steps: exact fixed point = %s  (%.0f s)" % (dt, rest, time.time() - t0), flush=True)
    g.close()
x = np.arange(N, dtype=np.float32)
wave = (0.02 * np.sin(2 * np.pi * x / N)).astype(np.float32)
g = luw.LBM(N, N, N, 0.01, fp16c=(dt == "fp16c"))
U = g.u.data.reshape(3, N, N, N)
U[0] = wave[None, :, None]                       # ux(y), uy(z), uz(x): every direction streams something different
U[1] = (0.5 * wave)[:, None, None]
U[2] = (0.25 * wave)[None, None, :]
g.run(12)
g.rho.read_from_device(); g.u.read_from_device()
mean = float(g.rho.data.astype(np.float64).mean()); fin = bool(np.isfinite(g.u.data).all()); umax = float(np.abs(g.u.data.reshape(3, -1)[0]).max())

dig_x = int(np.bitwise_xor.reduce(g.rho.data.view(np.uint32))) ^ int(np.bitwise_xor.reduce(g.u.data.view(np.uint32)))
dig_s = (int(g.rho.data.view(np.uint32).sum(dtype=np.uint64)) + int(g.u.data.view(np.uint32).sum(dtype=np.uint64))) & 0xFFFFFFFFFFFFFFFF
print("%s 1024^3 digest xor=%08x sum=%016x" % (dt, dig_x, dig_s), flush=True)
print("%s 1024^3 shear wave, 12 steps: mean rho - 1 = %.2e, finite = %s, max |ux| = %.5f (decaying from 0.02)  (%.0f s)" % (dt, mean - 1.0, fin, umax,
    time.time() - t0))
g.close()
assert rest and fin and abs(mean - 1.0) < (1e-4 if dt == "fp16c" else 1e-6) and 0.015 < umax <= 0.02
