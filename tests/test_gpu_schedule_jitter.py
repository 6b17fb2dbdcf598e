"""Schedule fuzzing of the multi-domain hosts (luw_dev_schedule_jitter): every second step / pack / unpack / edge kernel the library enqueues is held back
on its stream by a random delay of up to a few hundred microseconds -- far longer than the kernels of these small lattices.  Nothing may change: a result
that depends on one kernel being faster than another (a missing event between two streams or two domains) differs from the oracle under some seed.  The
negative control removes one real dependency (LUW_FAULT_UNPACK_WITHOUT_WAIT) and expects the fuzz to see it.  GPU only."""
import os

import numpy as np
import pytest

from helpers import synthetic_state, thermal_state

pytestmark = pytest.mark.gpu

SEEDS = (1, 2, 3, 4)
MAX_US = 300
MODES = {"one_phase": {}, "sequential": {"LUW_GROUP_EXCHANGE": "sequential"},
    "one_phase_threads": {"LUW_GROUP_THREADS": "1"}, "sequential_threads": {"LUW_GROUP_EXCHANGE": "sequential", "LUW_GROUP_THREADS": "1"},
    # the one-round exchange through send buffers: copies behind the pack kernels / ONE grouped ncclSend / ncclRecv batch per step (round 6)
    "one_phase_staged": {"LUW_GROUP_TRANSPORT": "staged"}, "one_phase_rccl": {"LUW_GROUP_TRANSPORT": "rccl"},
    # the alternatives first contact times against the defaults: x faces through the pack / insert kernels; the whole box as one launch, then the exchange
    "one_phase_x_packed": {"LUW_GROUP_EXCHANGE": "one_packed"}, "whole_box_then_exchange": {"LUW_GROUP_OVERLAP": "0"}}
KNOBS = ("LUW_GROUP_EXCHANGE", "LUW_GROUP_THREADS", "LUW_GROUP_TRANSPORT", "LUW_GROUP_OVERLAP")


def seeds_of(mode):
    """four seeds for the modes of round 5, two for the transports and alternatives added in round 6 (the suite's wall time; the 400-case fuzzer draws
    them at random: profiles/r06_exchange_fuzz.txt)"""
    return SEEDS if mode in ("one_phase", "sequential", "one_phase_threads", "sequential_threads") else SEEDS[:2]


@pytest.fixture(params=list(MODES))
def mode(request):
    from latticeurbanwind_amd import capi
    saved = {k: os.environ.get(k) for k in KNOBS}
    for k in KNOBS: os.environ.pop(k, None)
    os.environ.update(MODES[request.param])
    capi.reload_tuning()
    yield request.param
    capi.schedule_jitter(0, 0); capi.inject_fault(0)
    for k, v in saved.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    capi.reload_tuning()


def group(luw, gN, D, fp16c, st, **kw):
    g = luw.LBMGroup(*gN, *D, 0.01, fp16c=fp16c, devices=[0] * (D[0] * D[1] * D[2]), **kw)
    g.flags[:] = st[0]; g.u[:] = st[1]; g.rho[:] = st[2]
    return g


@pytest.mark.parametrize("gN,D,fp16c", [((24, 20, 16), (2, 2, 2), False), ((32, 24, 12), (4, 2, 1), False), ((640, 24, 16), (2, 1, 2), True),
    ((48, 20, 12), (3, 2, 1), True)])
def test_periodic_lattice_under_random_delays(luw, mode, gN, D, fp16c):
    from latticeurbanwind_amd import capi
    from oracle import oracle
    st = synthetic_state(*gN, seed=51, shell=None)
    o = oracle.OracleLBM(*gN, 0.01, fp16c=fp16c)
    o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
    o.run(9)
    for seed in seeds_of(mode):
        capi.schedule_jitter(seed, MAX_US)
        g = group(luw, gN, D, fp16c, st)
        g.run(0); g.run(5); g.run(4)
        g.read_from_device()
        capi.schedule_jitter(0, 0)
        assert np.array_equal(g.u, o.u) and np.array_equal(g.rho, o.rho), "seed %d" % seed
        g.close()


def test_forces_and_a_sampling_window_under_random_delays(luw, mode):
    """solid ground + TYPE_E shell, nudging + sponge + Coriolis, fused statistics in a sampling window (a sampled step's kernels take the x faces through the
    pack / unpack kernels: both ways of handing them over alternate here)"""
    from latticeurbanwind_amd import capi
    from oracle import oracle
    gN, D = (48, 40, 24), (2, 2, 1)
    st = synthetic_state(*gN, seed=53, shell="luw")
    o = oracle.OracleLBM(*gN, 0.01)
    o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
    o.set_coriolis(0.0, 3e-5, 4e-5); o.set_buffer_nudging(5, 0.0133333, 2, 1); o.set_top_sponge(6, 0.02)
    stats = oracle.OracleStats(o.N)
    o.run(3)
    for k in range(1, 10):
        o.run(1)
        if k >= 2 and (k - 2) % 3 == 0: stats.accumulate(o)
    for seed in seeds_of(mode):
        capi.schedule_jitter(seed, MAX_US)
        g = group(luw, gN, D, False, st, buffer_nudging=dict(n_cells=5, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1),
            top_sponge=dict(n_cells=6, inv_tau=0.02))
        g.set_coriolis(0.0, 3e-5, 4e-5)
        g.run(3); g.stats_reset(); g.run_sampled(9, 2, 3)
        g.read_from_device()
        d = g.stats_download()
        capi.schedule_jitter(0, 0)
        assert np.array_equal(g.u, o.u) and np.array_equal(g.rho, o.rho), "seed %d" % seed
        assert d["count"] == 3 and np.array_equal(d["avg_u"], stats.avg_u) and np.array_equal(d["m2_u"], stats.m2_u), "seed %d" % seed
        g.close()


def test_thermal_lattice_under_random_delays(luw, mode):
    from latticeurbanwind_amd import capi
    from oracle import oracle
    gN, D = (24, 20, 16), (2, 2, 2)
    st = synthetic_state(*gN, seed=55, shell=None)
    tflags, T = thermal_state(st[0], gN)
    o = oracle.OracleLBM(*gN, 0.01, fp16c=True, alpha=0.004)
    o.flags[:] = tflags; o.u[:] = st[1]; o.rho[:] = st[2]; o.T[:] = T
    o.run(6)
    for seed in seeds_of(mode):
        capi.schedule_jitter(seed, MAX_US)
        g = group(luw, gN, D, True, (tflags, st[1], st[2]), alpha=0.004)
        g.T[:] = T
        g.stats_reset(); g.run(0); g.run_sampled(6, 1, 2)
        g.read_from_device(("u", "rho", "T"))
        capi.schedule_jitter(0, 0)
        assert np.array_equal(g.u, o.u) and np.array_equal(g.T, o.T), "seed %d" % seed
        g.close()


@pytest.mark.parametrize("knobs", [{}, {"LUW_GROUP_EXCHANGE": "sequential"}])
def test_negative_control_a_missing_wait_is_seen(luw, knobs):
    """without the unpack kernels' wait for the neighbours' pack kernels the delays must produce a wrong result under at least one seed -- otherwise this
    file proves nothing"""
    from latticeurbanwind_amd import capi
    from oracle import oracle
    from latticeurbanwind_amd.capi import load
    saved = {k: os.environ.get(k) for k in KNOBS}
    for k in KNOBS: os.environ.pop(k, None)
    os.environ.update(knobs)
    capi.reload_tuning()
    try:
        gN, D = (32, 24, 12), (4, 2, 1)
        st = synthetic_state(*gN, seed=57, shell=None)
        o = oracle.OracleLBM(*gN, 0.01)
        o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
        o.run(9)
        seen = 0
        for seed in range(1, 9):
            capi.inject_fault(8)                   # LUW_FAULT_UNPACK_WITHOUT_WAIT
            capi.schedule_jitter(seed, MAX_US)
            g = group(luw, gN, D, False, st)
            g.run(0); g.run(9)
            g.read_from_device()
            capi.schedule_jitter(0, 0); capi.inject_fault(0)
            seen += not (np.array_equal(g.u, o.u) and np.array_equal(g.rho, o.rho))
            g.close()
        assert seen >= 1, "the schedule fuzz did not notice a missing dependency"
    finally:
        capi.schedule_jitter(0, 0); capi.inject_fault(0)
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        capi.reload_tuning()


@pytest.mark.parametrize("fp16c,own,D,thermal", [(False, (37, 9, 8), (2, 2, 1), False), (True, (256, 8, 7), (2, 2, 2), False),
    (True, (130, 7, 9), (2, 1, 2), True), (False, (20, 11, 6), (1, 2, 2), True)])
def test_the_per_rank_host_under_random_delays(luw, fp16c, own, D, thermal):
    """the one-process-per-GPU host (latticeurbanwind_amd/distributed.py: its schedule is luw_domain_step_launch plus the pack / exchange / unpack calls of
    hip_domain.py on the communication stream) as one rank that is its own neighbour: the one-round exchange under random delays against the three phases
    without them, rho, u, (T) and every DDF bit for bit"""
    from latticeurbanwind_amd import capi
    from latticeurbanwind_amd.distributed import DomainDecomposedLBM, DomainLayout, PeerLoopbackTransport
    from helpers import TYPE_S
    gN = tuple(o * d for o, d in zip(own, D))
    saved = {k: os.environ.get(k) for k in ("LUW_EXCHANGE", "LUW_X_SHELL")}
    os.environ["LUW_X_SHELL"] = "64"

    def run(exchange, seed):
        os.environ["LUW_EXCHANGE"] = exchange
        capi.schedule_jitter(seed, MAX_US if seed else 0)
        lay = DomainLayout(gN, D, 0)
        assert lay.can_overlap()
        sim = DomainDecomposedLBM(gN, D, 0.01, rank=0, transport=PeerLoopbackTransport(lay), fp16c=fp16c, device=0, **(dict(alpha=0.004) if thermal else {}))
        lx, ly, lz = lay.lN
        st = synthetic_state(lx, ly, lz, seed=59, shell=None)
        flags = st[0].reshape(lz, ly, lx).copy()
        flags[np.random.default_rng(59).random(flags.shape) < 0.04] = TYPE_S
        if thermal:
            tflags, T = thermal_state(flags.ravel(), (lx, ly, lz))
            sim.set_fields(tflags, st[1], st[2], T)
        else:
            sim.set_fields(flags.ravel(), st[1], st[2])
        sim.run(7)
        u, rho = sim.fields()
        out = [u.copy(), rho.copy(), np.asarray(sim.backend.lbm.download_fi()).copy()]
        if thermal: out += [sim.backend.download_T().copy(), np.asarray(sim.backend.lbm.download_gi()).copy()]
        sim.backend.close()
        capi.schedule_jitter(0, 0)
        return out
    try:
        ref = run("sequential", 0)
        for seed in SEEDS:
            got = run("batch", seed)
            assert all(np.array_equal(a, b) for a, b in zip(ref, got)), "seed %d" % seed
    finally:
        capi.schedule_jitter(0, 0)
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
