"""worker of tests/test_distributed_gloo.py: one rank of a gloo group, runs the product's multi-domain driver
(DomainDecomposedLBM + TorchDistTransport) over the oracle test double and ships its interior block to rank 0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
import torch
import torch.distributed as dist


def main():
    gNx, gNy, gNz, Dx, Dy, Dz, steps, fp16c, out = sys.argv[1:10]
    thermal = len(sys.argv) > 10 and sys.argv[10] == "thermal"
    gN = (int(gNx), int(gNy), int(gNz)); D = (int(Dx), int(Dy), int(Dz)); steps = int(steps); fp16c = bool(int(fp16c))
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    from latticeurbanwind_amd.distributed import DomainDecomposedLBM, DomainLayout
    from oracle_domain import OracleDomain
    from helpers import synthetic_state
    flags, u, rho = synthetic_state(*gN, seed=21, shell=None)           # fully periodic: exercises the wrap through the halo ring
    lay = DomainLayout(gN, D, rank)
    sim = DomainDecomposedLBM(gN, D, 0.01, rank=rank, backend=OracleDomain(lay, 0.01, fp16c=fp16c, alpha=0.004 if thermal else None))
    if thermal:
        from helpers import thermal_state
        tflags, T = thermal_state(flags, gN)
        sim.set_fields_from_global(tflags, u, rho, T)
    else:
        sim.set_fields_from_global(flags, u, rho)
    sim.run(steps)
    lu, lr = sim.fields()
    ub, off = sim.interior_to_global(lu, 3)
    rb, _ = sim.interior_to_global(sim.backend.download_T() if thermal else lr, 1)          # thermal runs ship T in the rho slot
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, (off, ub, rb))
    if rank == 0:
        U = np.zeros((3, gN[2], gN[1], gN[0]), np.float32); R = np.zeros((1, gN[2], gN[1], gN[0]), np.float32)
        for off, ub, rb in gathered:
            sl = (slice(None), slice(off[2], off[2] + ub.shape[1]), slice(off[1], off[1] + ub.shape[2]), slice(off[0], off[0] + ub.shape[3]))
            U[sl] = ub; R[sl] = rb
        np.savez(out, u=U.ravel(), rho=R.ravel())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
