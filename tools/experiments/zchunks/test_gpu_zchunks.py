"""The step pipelined along z (LUW_STEP_SCHEDULE=zchunks, latticeurbanwind_amd/zchunks.py): whole-row launches of z ranges, each range's faces travelling while
the following ranges are stepped.  One rank as its own neighbour (peer loopback: the buffers a step fills are the buffers the next one reads) against the
reference's three phases on the default schedule: rho, u and every DDF bit for bit -- with and without random delays in front of the library's kernels
(schedule fuzzing, tests/test_gpu_schedule_jitter.py).  The same schedule through RCCL's self send / receive: tests/test_gpu_bench_workloads.py and
tools/check_nccl_self.py under the same environment variable (profiles/r05_zchunks.txt).  GPU only."""
import os

import numpy as np
import pytest

from helpers import synthetic_state

pytestmark = pytest.mark.gpu


def run(luw, gN, D, fp16c, schedule, exchange, steps_list, jitter=0, forces=False):
    from latticeurbanwind_amd import capi
    from latticeurbanwind_amd.distributed import DomainDecomposedLBM, DomainLayout, PeerLoopbackTransport
    from helpers import TYPE_S
    saved = {k: os.environ.get(k) for k in ("LUW_EXCHANGE", "LUW_STEP_SCHEDULE", "LUW_X_SHELL")}
    os.environ["LUW_EXCHANGE"] = exchange; os.environ["LUW_X_SHELL"] = "64"
    if schedule: os.environ["LUW_STEP_SCHEDULE"] = schedule
    else: os.environ.pop("LUW_STEP_SCHEDULE", None)
    try:
        capi.schedule_jitter(jitter, 300 if jitter else 0)
        lay = DomainLayout(gN, D, 0)
        kw = {}
        if forces: kw = dict(buffer_nudging=dict(n_cells=3, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1), top_sponge=dict(n_cells=2, inv_tau=0.02))
        sim = DomainDecomposedLBM(gN, D, 0.01, rank=0, transport=PeerLoopbackTransport(lay), fp16c=fp16c, device=0, **kw)
        if forces:
            # nudging + sponge + Coriolis: the rank's share of a GLOBAL lattice with solid ground and TYPE_E faces, as in every LUW deck -- the reference cells
            # of the zone terms are inputs then (a FLUID reference cell is read and rewritten within one step: order-dependent in the reference itself)
            gst = synthetic_state(*gN, seed=61, shell="luw")
            gflags = gst[0].copy()
            gflags[(np.random.default_rng(61).random(gflags.shape) < 0.04) & ((gflags & 3) == 0)] = TYPE_S
            sim.set_fields_from_global(gflags, gst[1], gst[2])
            sim.backend.set_coriolis(0.0, 3e-5, 4e-5)
        else:
            lx, ly, lz = lay.lN
            st = synthetic_state(lx, ly, lz, seed=61, shell=None)
            flags = st[0].reshape(lz, ly, lx).copy()
            flags[np.random.default_rng(61).random(flags.shape) < 0.04] = TYPE_S
            sim.set_fields(flags.ravel(), st[1], st[2])
        from latticeurbanwind_amd import zchunks
        used = []
        for n in steps_list:
            used.append(zchunks.wanted() and zchunks.supported(sim, None))
            sim.run(n)
        u, rho = sim.fields()
        out = [u.copy(), rho.copy(), np.asarray(sim.backend.lbm.download_fi()).copy()]
        sim.backend.close()
        return out, used
    finally:
        capi.schedule_jitter(0, 0)
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


CASES = [(False, (37, 9, 16), (2, 2, 1), False), (False, (130, 8, 12), (2, 1, 1), False), (True, (256, 8, 9), (2, 2, 1), False),
    (True, (130, 7, 8), (1, 2, 1), False), (False, (40, 10, 13), (2, 2, 1), True), (True, (320, 6, 10), (2, 2, 1), True)]


@pytest.mark.parametrize("fp16c,own,D,forces", CASES)
def test_z_chunks_equal_the_three_phases_on_the_default_schedule(luw, fp16c, own, D, forces):
    gN = tuple(o * d for o, d in zip(own, D))
    ref, _ = run(luw, gN, D, fp16c, None, "sequential", (5, 4), forces=forces)
    got, used = run(luw, gN, D, fp16c, "zchunks", "batch", (5, 4), forces=forces)
    assert all(used), "the schedule under test did not run"
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))


@pytest.mark.parametrize("fp16c,own,D,forces", CASES[:4])
def test_z_chunks_under_random_delays(luw, fp16c, own, D, forces):
    gN = tuple(o * d for o, d in zip(own, D))
    ref, _ = run(luw, gN, D, fp16c, None, "sequential", (9,))
    for seed in (1, 2, 3, 4):
        got, used = run(luw, gN, D, fp16c, "zchunks", "batch", (9,), jitter=seed)
        assert all(used) and all(np.array_equal(a, b) for a, b in zip(ref, got)), "seed %d" % seed


def test_what_the_schedule_does_not_cover_takes_the_default_one(luw):
    """a z cut, too few layers, a sampled window: the default schedule runs, same values"""
    from latticeurbanwind_amd import zchunks
    ref, _ = run(luw, (40, 16, 12), (2, 2, 2), False, None, "sequential", (5,))
    got, used = run(luw, (40, 16, 12), (2, 2, 2), False, "zchunks", "batch", (5,))
    assert used == [False] and all(np.array_equal(a, b) for a, b in zip(ref, got))
    got, used = run(luw, (40, 16, 6), (2, 2, 1), False, "zchunks", "batch", (5,))
    assert used == [False]
    # FP16C rows of an odd number of cells take the one-cell kernel, which neither writes nor reads x faces itself: not for this schedule
    ref, _ = run(luw, (514, 14, 10), (2, 2, 1), True, None, "sequential", (3,))
    got, used = run(luw, (514, 14, 10), (2, 2, 1), True, "zchunks", "batch", (3,))
    assert used == [False] and all(np.array_equal(a, b) for a, b in zip(ref, got))
