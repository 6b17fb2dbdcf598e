"""worker of tests/test_gpu_bench_workloads.py: ONE rank of a multi-GPU benchmark tile in its real local shape, stepped through
the production path (DomainDecomposedLBM: shell on the communication stream, interior on the compute stream, pipelined steps,
halo faces through the real RCCL transport -- a one-rank world whose every neighbour is the rank itself) and compared bit for
bit with the CPU oracle stepping the SAME haloed domain (same decomposition parameters D, O; its own extract / swap / insert
following FX/lbm.cpp:1907-1935).  Physically: the rank's block made periodic.  tests/ only (imports the oracle).

usage: rank_shape_worker.py <f32|fp16c> <bx by bz> <Dx Dy Dz> <rank> <steps> [bld] [forcing] [cor] [peer] [switch]
(peer: PeerLoopbackTransport instead of RCCL -- the faces are written where the unpack reads them; switch: the run changes between the two step schedules,
through DomainDecomposedLBM.choose_schedule and by hand)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29655"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
    os.environ.setdefault(k, v)
import numpy as np
import torch
import torch.distributed as dist


def main():
    dt = sys.argv[1]
    block = tuple(int(v) for v in sys.argv[2:5]); D = tuple(int(v) for v in sys.argv[5:8]); rank = int(sys.argv[8]); steps = int(sys.argv[9])
    opts = set(sys.argv[10:])
    fp16c = dt == "fp16c"
    import latticeurbanwind_amd as luw
    from latticeurbanwind_amd.distributed import DomainDecomposedLBM, DomainLayout, SelfExchangeTransport as SelfNeighbour, init_rccl_process_group
    from bench import fill_channel, tile_forcing, coriolis_omega, NU
    from oracle import oracle
    from oracle_domain import OracleDomain
    torch.cuda.set_device(0)
    init_rccl_process_group(0)
    luw.load()
    gN = tuple(b * d for b, d in zip(block, D))
    nud, spg = tile_forcing() if "forcing" in opts else (None, None)

    lay = DomainLayout(gN, D, rank)
    if "peer" in opts:
        from latticeurbanwind_amd.distributed import PeerLoopbackTransport
        tr = PeerLoopbackTransport(lay)
    else:
        tr = SelfNeighbour(lay)
    tr.warm_up(torch.device("cuda", 0), torch.float16 if fp16c else torch.float32)      # connections first, lattice second (as in a real run)
    kw = dict(buffer_nudging=nud, top_sponge=spg) if nud else {}
    sim = DomainDecomposedLBM(gN, D, NU, rank=rank, transport=tr, fp16c=fp16c, device=0, **kw)
    assert sim.overlap, "the production schedule (shell / interior overlap) must be the one under test"
    lb = sim.backend.lbm
    fill_channel(lb.flags.data, lb.u.data, lb.rho.data, *lay.lN, *lay.O, *gN, buildings="bld" in opts)
    if "cor" in opts:
        sim.backend.set_coriolis(*coriolis_omega())

    od = OracleDomain(lay, NU, fp16c=fp16c)
    o = od.o
    o.flags[:] = lb.flags.data; o.u[:] = lb.u.data; o.rho[:] = lb.rho.data
    if nud:
        o.set_buffer_nudging(nud["n_cells"], nud["inv_tau"], nud["downstream_face"], nud["nudge_vertical"])
        o.set_top_sponge(spg["n_cells"], spg["inv_tau"])
    if "cor" in opts:
        o.set_coriolis(*coriolis_omega())

    def oracle_exchange():                           # communicate_fi with every neighbour = self: my + face lands in my - halo
        for a in lay.split_axes():
            bp, bm = o.extract_fi(a)
            o.insert_fi(a, bm, bp)                   # insert(recv_p, recv_m): recv_p = what the + neighbour sent downwards = my bm

    sim.initialize()
    o.initialize(); o.t = 1; oracle_exchange(); o.t = 0      # FX/lbm.cpp:1242-1258
    if "switch" in opts:
        # the start-up probe of bench.py --gpus N (choose_schedule: real steps under each schedule, the state simply advances), then both schedules once more
        # by hand: every switch between "shell first" and "whole box" must leave the run on the oracle's values
        pr = sim.choose_schedule(steps=2)
        assert pr and pr["shell_first_ms"] > 0 and pr["whole_box_ms"] > 0 and pr["kept"], pr
        assert sim.set_schedule(False) is False
        sim.run(2)
        assert sim.set_schedule(True) is True
        sim.run(steps)
        steps += 2 * (3 + 2) + 2
    else:
        sim.run(steps)
    for _ in range(steps):
        o.stream_collide(); oracle_exchange(); o.t += 1
    gu, grho = sim.fields()
    fi = lb.download_fi()
    own = tuple(slice(h, n - h) for h, n in zip(lay.H, lay.lN))[::-1]                 # (z, y, x)
    cut = lambda a, c: np.asarray(a).reshape((c,) + tuple(lay.lN[::-1]))[(slice(None),) + own]
    ok_u = np.array_equal(cut(gu, 3), cut(o.u, 3)); ok_r = np.array_equal(cut(grho, 1), cut(o.rho, 1))
    a, b = cut(fi, 19), cut(o.fi, 19)
    if fp16c:
        ok_f = all(np.array_equal(np.where(a[i] == 0x8000, 0, a[i]), np.where(b[i] == 0x8000, 0, b[i])) for i in range(19))
    else:
        ok_f = all(np.array_equal(a[i], b[i]) for i in range(19))
    moved = float(np.abs(cut(gu, 3)[1]).max())       # the flow must have developed a cross-wind component somewhere: not a trivial state
    print("rank-shape %s local %s of n_gpu %s rank %d, %d steps, opts %s: u equal %s, rho equal %s, DDFs equal %s, max|uy| %.3e" % (dt, lay.lN, D, rank, steps,
        sorted(opts), ok_u, ok_r, ok_f, moved))
    sim.backend.close()
    dist.destroy_process_group()
    assert ok_u and ok_r and ok_f


if __name__ == "__main__":
    main()
