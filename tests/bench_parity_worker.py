"""worker of tests/test_bench_selfcheck.py: one rank of a gloo group runs bench.py's N > 1 self-check logic (owned-cell digests of a
decomposed urban tile, all-gathered, against the undivided oracle's) with the oracle test double as the domain.  tests/ only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
import torch.distributed as dist


def main():
    gN = tuple(int(v) for v in sys.argv[1:4]); D = tuple(int(v) for v in sys.argv[4:7]); fp16c = bool(int(sys.argv[7])); corrupt = int(
        sys.argv[8]); out = sys.argv[9]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import bench
    from benchmarks import multi
    from latticeurbanwind_amd.distributed import DomainDecomposedLBM, DomainLayout
    from oracle_domain import OracleDomain
    nud, spg = bench.tile_forcing()
    forcing = (dict(nud, n_cells=3), dict(spg, n_cells=4))        # zones thinner than a rank's block
    lay = DomainLayout(gN, D, rank)
    od = OracleDomain(lay, bench.NU, fp16c=fp16c)
    od.o.set_buffer_nudging(forcing[0]["n_cells"], forcing[0]["inv_tau"], forcing[0]["downstream_face"], forcing[0]["nudge_vertical"])
    od.o.set_top_sponge(forcing[1]["n_cells"], forcing[1]["inv_tau"])
    od.o.set_coriolis(*bench.coriolis_omega())
    sim = DomainDecomposedLBM(gN, D, bench.NU, rank=rank, backend=od)
    bench.fill_channel(od.o.flags, od.o.u, od.o.rho, *lay.lN, *lay.O, *gN, buildings=True)
    sim.run(4)
    u, rho = sim.fields()
    if corrupt and rank == world - 1:
        # one owned value of one rank: the check must see it
        u = u.copy(); u[((lay.lN[2] // 2) * lay.lN[1] + lay.lN[1] // 2) * lay.lN[0] + lay.lN[0] // 2] += np.float32(1e-6)
    mine = multi.owned_digests(lay, u, rho, od.o.fi, fp16c)
    got = [None] * world
    dist.all_gather_object(got, mine)
    if rank == 0:
        o = multi.oracle_tile(gN, fp16c, True, 4, forcing)
        want = multi.oracle_digests(o, gN, D, world, fp16c)
        bad = [[f for f in ("rho", "u", "fi") if got[r][f] != want[r][f]] for r in range(world)]
        json.dump({"bad": bad, "max_abs_uy": max(g["max_abs_uy"] for g in got)}, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
