#!/usr/bin/env python3
"""The production halo transport (TorchDistTransport: RCCL batch_isend_irecv on the communication stream, FP16C codes travelling
as float16) exercised on ONE GPU: a world of one rank whose every neighbour is the rank itself, so each face is sent to and
received from the same process through RCCL's self send/recv -- physically the periodic single-domain problem, the same as the
in-process loopback used by bench_domain_overhead.py.  Both runs must leave identical bits; prints the step times.
usage (GPU box):  MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python3 tools/check_nccl_self.py [f32|fp16c] [sx sy sz Dx Dy Dz]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29641"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
    os.environ.setdefault(k, v)
import numpy as np
import torch
import torch.distributed as dist
import latticeurbanwind_amd as luw
from latticeurbanwind_amd.distributed import DomainDecomposedLBM, SelfExchangeTransport as SelfNeighbour, init_rccl_process_group
from bench import channel_state
from tools.bench_domain_overhead import Loopback

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
torch.cuda.set_device(0)
init_rccl_process_group(0)
luw.load()
if os.environ.get("LUW_SELF_EARLY", "1") == "1":        # as DomainDecomposedLBM does in a real run: RCCL connections before any lattice exists (0: A/B aid)
    x, y = torch.ones(1024, device="cuda"), torch.zeros(1024, device="cuda")
    for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, x, 0), dist.P2POp(dist.irecv, y, 0)]): r.wait()
    torch.cuda.synchronize()
D, size, steps = (1, 2, 2), (512, 256, 256), int(os.environ.get("LUW_SELF_STEPS", "60"))
if len(sys.argv) >= 8:      # check_nccl_self.py f32 2048 256 256 1 4 2: one rank of the 8-GPU benchmark tile
    size, D = tuple(int(v) for v in sys.argv[2:5]), tuple(int(v) for v in sys.argv[5:8])
N = tuple(s * d for s, d in zip(size, D))


def run(transport_of):
    sim = DomainDecomposedLBM(N, D, 1.48e-7, rank=0, transport=None if transport_of is None else Loopback(),
        overlap=(os.environ.get("LUW_SELF_OVERLAP", "1") != "0"), fp16c=(dt == "fp16c"), device=0)
    if transport_of is not None:
        sim.transport = transport_of(sim.layout)
    fl, u, rho = channel_state(sim.lNx, sim.lNy, sim.lNz, *sim.global_offset, *N)
    sim.set_fields(fl, u, rho); sim.initialize(); sim.run(5)
    torch.cuda.synchronize(); t0 = time.perf_counter(); k_ms = (sim.run(steps, timed=True) or {}).get("kernel_ms"); torch.cuda.synchronize(); ms = (
        time.perf_counter() - t0) / steps * 1e3
    sys.stderr.write("  %s: %.3f ms/step, interior kernel %.3f ms\n" % (type(sim.transport).__name__, ms, k_ms or 0.0))
    uu, rr = sim.fields()
    out = (uu.copy(), rr.copy(), ms)
    sim.backend.close()
    return out


def run_undivided():
    """the same physical problem without any decomposition: the owned block of rank 0 as ONE periodic lattice (whole-box kernel,
    no halos, no second stream) -- what shell/interior overlap, pipelined steps and the exchange must reproduce bit for bit"""
    from latticeurbanwind_amd.lbm import LBM
    from latticeurbanwind_amd.distributed import DomainLayout
    lay = DomainLayout(N, D, 0)
    fl, u, rho = channel_state(*lay.lN, *lay.O, *N)
    own = tuple(slice(h, n - h) for h, n in zip(lay.H, lay.lN))[::-1]                 # (z, y, x)
    cut = lambda a, c: np.ascontiguousarray(np.asarray(a).reshape((c,) + tuple(lay.lN[::-1]))[(slice(None),) + own]).ravel()
    lbm = LBM(*size, 1.48e-7, fp16c=(dt == "fp16c"), device=0)
    lbm.flags.data[:] = cut(fl, 1); lbm.u.data[:] = cut(u, 3); lbm.rho.data[:] = cut(rho, 1)
    lbm.run(0); lbm.run(5); lbm.run(steps)
    lbm.u.read_from_device(); lbm.rho.read_from_device()
    out = (lbm.u.data.copy(), lbm.rho.data.copy(), cut)
    lbm.close()
    return out


if os.environ.get("LUW_SELF_ORDER", "ab") == "ba":       # A/B aid: which run comes first in the process
    b = run(lambda lay: SelfNeighbour(lay)); a = run(lambda lay: Loopback())
else:
    a = run(lambda lay: Loopback()); b = run(lambda lay: SelfNeighbour(lay))
c = run_undivided()
same = np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
undiv = np.array_equal(c[2](b[0], 3).view(np.uint32), c[0].view(np.uint32)) and np.array_equal(c[2](b[1], 1).view(np.uint32), c[1].view(np.uint32))
print("%s D=%s local %s: loopback %.3f ms/step, RCCL self send/recv %.3f ms/step, fields identical: %s, equal to the undivided periodic run: %s" % (dt, D, size,
    a[2], b[2], same, undiv))
dist.destroy_process_group()
assert same and undiv
