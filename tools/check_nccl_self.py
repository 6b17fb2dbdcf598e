#!/usr/bin/env python3
"""The production halo transport (TorchDistTransport: RCCL batch_isend_irecv on the communication stream, FP16C codes travelling
as float16) exercised on ONE GPU: a world of one rank whose every neighbour is the rank itself, so each face is sent to and
received from the same process through RCCL's self send/recv -- physically the periodic single-domain problem, the same as the
in-process loopback used by bench_domain_overhead.py.  Both runs must leave identical bits; prints the step times.
usage (GPU box):  MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python3 tools/check_nccl_self.py [f32|fp16c]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29641"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
    os.environ.setdefault(k, v)
import numpy as np
import torch
import torch.distributed as dist
import latticeurbanwind_amd as luw
from latticeurbanwind_amd.distributed import DomainDecomposedLBM, TorchDistTransport
from bench import channel_state
from tools.bench_domain_overhead import Loopback

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
luw.load()
D, size, steps = (1, 2, 2), (512, 256, 256), 60
N = tuple(s * d for s, d in zip(size, D))


class SelfNeighbour(TorchDistTransport):
    def __init__(self, layout):
        super().__init__(layout)
        self.layout = type("L", (), {"neighbor": staticmethod(lambda axis, sign: 0)})()


def run(transport_of):
    sim = DomainDecomposedLBM(N, D, 1.48e-7, rank=0, transport=None if transport_of is None else Loopback(), overlap=True, fp16c=(dt == "fp16c"), device=0)
    if transport_of is not None:
        sim.transport = transport_of(sim.layout)
    fl, u, rho = channel_state(sim.lNx, sim.lNy, sim.lNz, *sim.global_offset, *N)
    sim.set_fields(fl, u, rho); sim.initialize(); sim.run(5)
    torch.cuda.synchronize(); t0 = time.perf_counter(); sim.run(steps); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    uu, rr = sim.fields()
    out = (uu.copy(), rr.copy(), ms)
    sim.backend.close()
    return out


a = run(lambda lay: Loopback())
b = run(lambda lay: SelfNeighbour(lay))
same = np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
print("%s D=%s local %s: loopback %.3f ms/step, RCCL self send/recv %.3f ms/step, fields identical: %s" % (dt, D, size, a[2], b[2], same))
dist.destroy_process_group()
assert same
