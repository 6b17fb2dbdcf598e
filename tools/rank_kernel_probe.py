#!/usr/bin/env python3
"""One rank of the 2048x1024x512 urban tile on one GPU: the interior-box kernel ALONE (nothing else on the device) against the same kernel inside the
production step (boundary shell + pack / RCCL self exchange / unpack running beside it on the high-priority communication stream).
usage: rank_kernel_probe.py <f32|fp16c> <Dx Dy Dz> <rank> [cor]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29647"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
    os.environ.setdefault(k, v)
import torch
import latticeurbanwind_amd as luw
from latticeurbanwind_amd.distributed import DomainDecomposedLBM, DomainLayout, SelfExchangeTransport, init_rccl_process_group
from bench import fill_channel, tile_forcing, coriolis_omega, NU
fp16c = sys.argv[1] == "fp16c"; D = tuple(int(v) for v in sys.argv[2:5]); rank = int(sys.argv[5]); cor = "cor" in sys.argv[6:]
torch.cuda.set_device(0); init_rccl_process_group(0); luw.load()
gN = (2048, 1024, 512)          # BASELINE configs[3] / configs[4]; D must divide it
nud, spg = tile_forcing()
lay = DomainLayout(gN, D, rank)
tr = SelfExchangeTransport(lay); tr.warm_up(torch.device("cuda", 0), torch.float16 if fp16c else torch.float32)
sim = DomainDecomposedLBM(gN, D, NU, rank=rank, transport=tr, fp16c=fp16c, device=0, buffer_nudging=nud, top_sponge=spg)
lb = sim.backend.lbm
fill_channel(lb.flags.data, lb.u.data, lb.rho.data, *sim.layout.lN, *sim.layout.O, *gN, buildings=True)
if cor: sim.backend.set_coriolis(*coriolis_omega())
sim.initialize(); sim.run(10)
b = sim.backend
def alone(box, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(b.compute)
    for _ in range(n): b.stream_collide(box, 0, b.compute); b.increment_time_step(1)
    e1.record(b.compute); e1.synchronize()
    return e0.elapsed_time(e1) / n
print("local", sim.layout.lN, "interior", sim.layout.interior_box(), "shell boxes", sim.layout.shell_boxes())
print("interior kernel alone      %.3f ms" % alone(sim.layout.interior_box()))
print("whole-box kernel alone     %.3f ms" % alone(sim.layout.whole_box()))
for bx in sim.layout.shell_boxes(): print("shell box %s alone %.3f ms" % (bx, alone(bx)))
t0 = time.perf_counter(); tm = sim.run(60, timed=True); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 60 * 1e3
print("production step            %.3f ms wall; interior kernel inside it %.3f ms, shell %.3f ms, pack+exchange+unpack %.3f ms" % (ms, tm["kernel_ms"],
    tm["shell_ms"] or 0.0, tm["exchange_ms"]))
sim.backend.close(); torch.distributed.destroy_process_group()
