"""The multi-domain host runtime of the C-ABI (luw_group_*: the reference's `LBM(N, Dx, Dy, Dz, ...)` in ONE process) on the test
box's single GPU: 2..8 domains share device 0, every face travels between the domains' buffers inside the library (peer
stores of the pack kernels, or the staged copy path), shell / interior overlap and pipelined steps included.  The global
fields must equal the oracle's run of the undivided lattice bit for bit.  GPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import synthetic_state, thermal_state

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # global lattice, n_gpu, FP16C, expect shell / interior overlap
    ((24, 20, 16), (2, 1, 1), False, True), ((24, 20, 16), (2, 2, 2), False, True), ((32, 24, 12), (4, 2, 1), False, True),
        ((26, 18, 16), (1, 3, 2), True, True),
    ((16, 12, 6), (2, 2, 2), False, False),             # 3 owned layers in z: too thin for a shell, whole box + exchange
    ((640, 24, 16), (2, 1, 2), True, True),             # rows wide enough for the FP16C pair kernel and 64-cell x slabs
    ((260, 12, 12), (1, 2, 2), False, True),
]


@pytest.fixture(params=["one_phase", "sequential"])
def exchange(request):
    """both exchange routes of luw_group_*: ONE pack / unpack round per step (default with peer stores: faces of all axes, twelve edge messages, x faces read
    in place by the next step's kernels) and the reference's three phases x, y, z (LUW_GROUP_EXCHANGE=sequential)"""
    from latticeurbanwind_amd import capi
    saved = {k: os.environ.get(k) for k in ("LUW_GROUP_EXCHANGE",)}
    for k in saved: os.environ.pop(k, None)
    if request.param == "sequential": os.environ["LUW_GROUP_EXCHANGE"] = "sequential"
    capi.reload_tuning()
    yield request.param
    for k, v in saved.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    capi.reload_tuning()


def run_group(luw, gN, D, fp16c, state, steps, **kw):
    n = D[0] * D[1] * D[2]
    g = luw.LBMGroup(*gN, *D, 0.01, fp16c=fp16c, devices=[0] * n, **kw)
    g.flags[:] = state[0]; g.u[:] = state[1]; g.rho[:] = state[2]
    return g


@pytest.mark.parametrize("gN,D,fp16c,overlap", CASES)
def test_group_equals_oracle_on_the_undivided_lattice(luw, gN, D, fp16c, overlap, exchange):
    from oracle import oracle
    st = synthetic_state(*gN, seed=41, shell=None)                      # fully periodic: the wrap runs through the halo ring
    g = run_group(luw, gN, D, fp16c, st, 7)
    assert g.overlaps() == overlap and g.direct_peer_stores() and g.one_phase() == (exchange != "sequential")
    g.run(0); g.run(4); g.run(3)                                        # two calls: events of the first are reused by the second
    g.read_from_device()
    o = oracle.OracleLBM(*gN, 0.01, fp16c=fp16c)
    o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
    o.run(7)
    assert g.get_t() == 7
    assert np.array_equal(g.u, o.u) and np.array_equal(g.rho, o.rho)
    g.close()


def test_group_with_luw_shell_forces_and_statistics(luw, exchange):
    """solid ground + TYPE_E shell, nudging + sponge + Coriolis, a sampling window: fields and Welford statistics of a
    [2,2,1] group against the oracle (+ host Welford) on the undivided lattice"""
    from oracle import oracle
    gN, D = (48, 40, 24), (2, 2, 1)
    st = synthetic_state(*gN, seed=43, shell="luw")
    nud = dict(n_cells=5, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1); spg = dict(n_cells=6, inv_tau=0.02)
    g = run_group(luw, gN, D, False, st, 0, buffer_nudging=nud, top_sponge=spg)
    g.set_coriolis(0.0, 3e-5, 4e-5)
    o = oracle.OracleLBM(*gN, 0.01)
    o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
    o.set_coriolis(0.0, 3e-5, 4e-5); o.set_buffer_nudging(5, 0.0133333, 2, 1); o.set_top_sponge(6, 0.02)
    g.run(5); o.run(5)
    g.stats_reset()
    stats = oracle.OracleStats(o.N)
    g.run_sampled(9, 2, 3)                                             # samples at steps 2, 5, 8 of the window
    for k in range(1, 10):
        o.run(1)
        if k >= 2 and (k - 2) % 3 == 0:
            stats.accumulate(o)
    g.read_from_device()
    assert np.array_equal(g.u, o.u) and np.array_equal(g.rho, o.rho)
    d = g.stats_download()
    assert d["count"] == 3 == stats.count
    assert np.array_equal(d["avg_u"], stats.avg_u) and np.array_equal(d["avg_rho"], stats.avg_rho)     # avg_u: AoS [3n+c] on both sides
    assert np.array_equal(d["m2_u"], stats.m2_u) and np.array_equal(d["m2_v"], stats.m2_v) and np.array_equal(d["m2_w"], stats.m2_w)
    g.close()


@pytest.mark.parametrize("D,fp16c", [((2, 1, 1), False), ((2, 2, 2), True)])
def test_group_thermal_lattice(luw, D, fp16c, exchange):
    from oracle import oracle
    gN = (24, 20, 16)
    st = synthetic_state(*gN, seed=45, shell=None)
    tflags, T = thermal_state(st[0], gN)
    g = run_group(luw, gN, D, fp16c, (tflags, st[1], st[2]), 0, alpha=0.004)
    g.T[:] = T
    g.stats_reset()
    g.run(0)
    g.run_sampled(6, 1, 2)                                             # thermal lattice: separate statistics kernel between pipelined steps
    g.read_from_device(("u", "rho", "T"))
    o = oracle.OracleLBM(*gN, 0.01, fp16c=fp16c, alpha=0.004)
    o.flags[:] = tflags; o.u[:] = st[1]; o.rho[:] = st[2]; o.T[:] = T
    o.run(6)
    assert np.array_equal(g.u, o.u) and np.array_equal(g.T, o.T) and o.T.std() > 1e-4
    assert g.stats_download()["count"] == 3
    g.close()


def test_group_voxelise_gather_and_inlet_match_the_single_domain(luw):
    """per-domain voxelisation, probe gather and the von-Karman inlet split over the owners: a [2,2,1] group equals ONE domain"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_refcases import box_tris
    gN, D = (48, 40, 24), (2, 2, 1)
    st = synthetic_state(*gN, seed=47, solids=False, shell="luw")
    # (T, 3, 3); the second box straddles the domain cut
    tri = np.array(box_tris(10.0, 22.0, 8.0, 19.0, 1.0, 9.0) + box_tris(21.0, 30.0, 18.0, 27.0, 1.0, 14.0), np.float32)
    bounds = np.concatenate([tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0)])
    rng = np.random.default_rng(3)
    Nx, Ny, Nz = gN
    zs, ys = np.meshgrid(np.arange(1, Nz - 1), np.arange(Ny), indexing="ij")
    cells = (0 + (ys.ravel() + zs.ravel() * Ny) * Nx).astype(np.uint64)            # the west face, both y halves
    P, M = cells.size, 8
    pdata = np.zeros((7, P), np.float32); pdata[0] = -0.5 * Nx + 0.5; pdata[1] = ys.ravel() - 0.5 * Ny + 0.5; pdata[2] = zs.ravel() - 0.5 * Nz + 0.5
    pdata[3] = 0.05; pdata[6] = 0.01
    mdata = (0.3 * rng.standard_normal((10, 5 * M))).astype(np.float32)
    probes = np.array([5 + (7 + 3 * Ny) * Nx, 40 + (30 + 5 * Ny) * Nx, 30 + (10 + 20 * Ny) * Nx], np.uint64)
    res = []
    for kind in ("single", "group"):
        g = luw.LBM(*gN, 0.01) if kind == "single" else luw.LBMGroup(*gN, *D, 0.01, devices=[0] * 4)
        fl, u, rho = (g.flags.data, g.u.data, g.rho.data) if kind == "single" else (g.flags, g.u, g.rho)
        fl[:] = st[0]; u[:] = st[1]; rho[:] = st[2]
        g.voxelize_mesh_on_device(tri, bounds=bounds)
        mask = fl.copy()
        g.vk_inlet_attach(cells, np.zeros(P, np.uint8), pdata.ravel(), mdata.ravel(), M)
        g.gather_attach(probes)
        g.run(6)
        pu = g.gather_u()
        if kind == "single":
            g.u.read_from_device(); g.rho.read_from_device()
            res.append((mask, g.u.data.copy(), g.rho.data.copy(), pu))
        else:
            g.read_from_device()
            res.append((mask, g.u.copy(), g.rho.copy(), pu))
        g.close()
    assert (res[0][0] & 1).sum() > 1000
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("exchange_env", ["", "sequential"])
def test_group_staged_copy_path(tmp_path, exchange_env):
    """the path for devices without peer access (pack into a send buffer, hipMemcpyPeerAsync into the neighbour's receive
    buffer), forced with LUW_GROUP_TRANSPORT=staged in a child process -- in ONE round per step (faces of all axes, edge messages, x faces written by the
    step kernels and read in place; round 6) and in the reference's three phases: same bits as the oracle, thermal lattice and an x-split FP16C case included"""
    code = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import latticeurbanwind_amd as luw
from helpers import synthetic_state, thermal_state
from oracle import oracle
luw.load()
for gN, D, fp16c, alpha in (((32, 24, 12), (2, 2, 1), False, None), ((640, 24, 16), (2, 1, 2), True, None), ((24, 20, 16), (2, 2, 2), False, 0.004)):
    st = synthetic_state(*gN, seed=49, shell=None)
    fl, T = thermal_state(st[0], gN) if alpha else (st[0], None)
    g = luw.LBMGroup(*gN, *D, 0.01, fp16c=fp16c, devices=[0] * (D[0] * D[1] * D[2]), **({"alpha": alpha} if alpha else {}))
    assert not g.direct_peer_stores() and g.one_phase() == (%r != "sequential")
    g.flags[:] = fl; g.u[:] = st[1]; g.rho[:] = st[2]
    if alpha: g.T[:] = T
    g.run(0); g.run(4); g.run(3); g.read_from_device(("u", "rho", "T") if alpha else ("u", "rho"))
    o = oracle.OracleLBM(*gN, 0.01, fp16c=fp16c, alpha=alpha); o.flags[:] = fl; o.u[:] = st[1]; o.rho[:] = st[2]
    if alpha: o.T[:] = T
    o.run(7)
    assert np.array_equal(g.u, o.u) and np.array_equal(g.rho, o.rho) and (not alpha or np.array_equal(g.T, o.T)), (gN, D)
    g.close()
print("staged ok")
""" % (ROOT, os.path.join(ROOT, "tests"), exchange_env)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LUW_GROUP_TRANSPORT="staged", LUW_GROUP_EXCHANGE=exchange_env), capture_output=True,
        text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "staged ok" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


@pytest.mark.parametrize("exchange_env", ["", "sequential"])
def test_group_rccl_transport_self(luw, exchange_env):
    """LUW_GROUP_TRANSPORT=rccl (grouped ncclSend / ncclRecv, librccl loaded on demand): with all domains on the box's one GPU the
    communicator has one rank and every message is a self send / receive -- message pairing, buffer roles and stream ordering of the
    transport are the ones a node uses.  ONE batch per step (faces of all axes, twelve edge messages, thermal faces; round 6) and the reference's three
    phases: same bits as the oracle, thermal lattice included"""
    from latticeurbanwind_amd import capi
    from oracle import oracle
    saved = {k: os.environ.get(k) for k in ("LUW_GROUP_TRANSPORT", "LUW_GROUP_EXCHANGE")}
    os.environ["LUW_GROUP_TRANSPORT"] = "rccl"; os.environ["LUW_GROUP_EXCHANGE"] = exchange_env
    capi.reload_tuning()    # the library reads its environment once: read it again
    try:
        for gN, D, fp16c, alpha in (((32, 24, 12), (2, 2, 1), False, None), ((640, 24, 16), (2, 1, 2), True, None), ((24, 20, 16), (2, 2, 2), False, 0.004)):
            st = synthetic_state(*gN, seed=53, shell=None)
            tflags, T = thermal_state(st[0], gN) if alpha else (st[0], None)
            g = run_group(luw, gN, D, fp16c, (tflags, st[1], st[2]), 0, **({"alpha": alpha} if alpha else {}))
            assert g.transport() == 2 and not g.direct_peer_stores() and g.one_phase() == (exchange_env != "sequential"), capi.TRANSPORT_NAMES[g.transport()]
            if alpha: g.T[:] = T
            g.run(0); g.run(4); g.run(3)
            g.read_from_device(("u", "rho", "T") if alpha else ("u", "rho"))
            o = oracle.OracleLBM(*gN, 0.01, fp16c=fp16c, alpha=alpha)
            o.flags[:] = tflags; o.u[:] = st[1]; o.rho[:] = st[2]
            if alpha: o.T[:] = T
            o.run(7)
            assert np.array_equal(g.u, o.u) and np.array_equal(g.rho, o.rho) and (not alpha or np.array_equal(g.T, o.T))
            g.close()
    finally:
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        capi.reload_tuning()


@pytest.mark.parametrize("transport", ["peer", "staged", "rccl"])
@pytest.mark.parametrize("threads", ["0", "1"])
def test_group_on_distinct_devices(luw, transport, threads):
    """the wire: domains on DIFFERENT GPUs (default device list: one device per domain) -- peer access, remote stores of the pack kernel
    into another GPU's receive buffer, cross-device stream waits, hipMemcpyPeerAsync, multi-rank RCCL.  Skipped on the one-GPU test box;
    the same check runs inside `bench.py --gpus N` (secondary.group_host) whenever a node is available."""
    import torch
    from oracle import oracle
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("needs at least two GPUs")
    if transport == "rccl" and threads == "1":
        pytest.skip("RCCL's grouped calls are issued by one thread")
    D = (2, 2, 1) if ndev >= 4 else (2, 1, 1)
    gN = (640, 48, 24)
    saved = {k: os.environ.get(k) for k in ("LUW_GROUP_TRANSPORT", "LUW_GROUP_THREADS")}
    os.environ.update(LUW_GROUP_TRANSPORT=transport, LUW_GROUP_THREADS=threads)
    from latticeurbanwind_amd import capi
    capi.reload_tuning()
    try:
        for fp16c in (False, True):
            st = synthetic_state(*gN, seed=55, shell="luw")
            g = luw.LBMGroup(*gN, *D, 0.01, fp16c=fp16c)                 # devices=None: domain d on device d
            assert len({g.domain_info(d)[2] for d in range(g.get_D())}) == g.get_D()
            g.flags[:] = st[0]; g.u[:] = st[1]; g.rho[:] = st[2]
            g.run(0); g.run(6); g.run(5)
            g.read_from_device()
            o = oracle.OracleLBM(*gN, 0.01, fp16c=fp16c)
            o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
            o.run(11)
            assert np.array_equal(g.u, o.u) and np.array_equal(g.rho, o.rho), (transport, threads, fp16c)
            g.close()
    finally:
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        capi.reload_tuning()


def test_group_one_host_thread_per_domain(tmp_path):
    """LUW_GROUP_THREADS=1: every domain is enqueued by its own host thread (calls of four steps and more; the default is one thread for
    all), neighbours ordered through published exchange numbers: same bits as the oracle, sampled window and thermal lattice included"""
    code = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import latticeurbanwind_amd as luw
from helpers import synthetic_state, thermal_state
from oracle import oracle
luw.load()
gN, D = (32, 24, 12), (2, 2, 1)
st = synthetic_state(*gN, seed=51, shell=None)
tflags, T = thermal_state(st[0], gN)
g = luw.LBMGroup(*gN, *D, 0.01, devices=[0] * 4, alpha=0.004)
g.flags[:] = tflags; g.u[:] = st[1]; g.rho[:] = st[2]; g.T[:] = T
g.stats_reset(); g.run(0); g.run_sampled(7, 2, 2); g.run(2); g.read_from_device(("u", "rho", "T"))
o = oracle.OracleLBM(*gN, 0.01, alpha=0.004); o.flags[:] = tflags; o.u[:] = st[1]; o.rho[:] = st[2]; o.T[:] = T; o.run(9)
assert np.array_equal(g.u, o.u) and np.array_equal(g.rho, o.rho) and np.array_equal(g.T, o.T) and g.stats_download()["count"] == 3
print("threads ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LUW_GROUP_THREADS="1"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "threads ok" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


def test_group_refuses_what_the_reference_refuses(luw):
    from latticeurbanwind_amd import capi
    with pytest.raises(capi.LuwError):
        luw.LBMGroup(25, 20, 16, 2, 1, 1, 0.01, devices=[0, 0])          # not divisible: the caller shrinks the grid first
    with pytest.raises(capi.LuwError):
        luw.LBMGroup(24, 20, 16, 2, 1, 1, 0.01)                          # one device per domain unless a device list says otherwise (FX/lbm.cpp:961-979)


def test_pairs_without_peer_access_fall_back_to_copies_pair_by_pair(luw):
    """first-contact insurance (luw_dev_inject_fault): half of the domain pairs have "no peer access" -- their faces go through send buffers and
    hipMemcpyPeerAsync while the other pairs keep the pack kernels' peer stores, in the same exchange; same bits as the oracle, overlap and pipelining on"""
    from latticeurbanwind_amd import capi
    from oracle import oracle
    capi.inject_fault(capi.FAULT_NO_PEER_ODD_PAIRS)
    try:
        for gN, D, fp16c in (((32, 24, 12), (4, 2, 1), False), ((640, 24, 16), (2, 1, 2), True)):
            st = synthetic_state(*gN, seed=47, shell=None)
            g = run_group(luw, gN, D, fp16c, st, 0)
            assert g.overlaps() and not g.direct_peer_stores() and g.transport() == 1     # reported as staged: some pair has no peer access
            assert not g.one_phase()                                                       # ... and the exchange falls back to the three phases
            g.run(0); g.run(5); g.run(4); g.read_from_device()
            o = oracle.OracleLBM(*gN, 0.01, fp16c=fp16c)
            o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
            o.run(9)
            assert np.array_equal(g.u, o.u) and np.array_equal(g.rho, o.rho)
            g.close()
    finally:
        capi.inject_fault(0)


def test_failing_rccl_initialisation_is_a_clean_error(luw):
    """LUW_GROUP_TRANSPORT=rccl with ncclCommInitAll failing (injected): luw_group_create returns an error with a message, nothing hangs, nothing leaks
    into the next group, which runs on peer stores as if nothing had happened"""
    from latticeurbanwind_amd import capi
    from oracle import oracle
    saved = os.environ.get("LUW_GROUP_TRANSPORT")
    os.environ["LUW_GROUP_TRANSPORT"] = "rccl"; capi.reload_tuning(); capi.inject_fault(capi.FAULT_RCCL_INIT)
    gN, D = (24, 20, 16), (2, 2, 1)
    st = synthetic_state(*gN, seed=49, shell=None)
    try:
        with pytest.raises(capi.LuwError, match="ncclCommInitAll"):
            run_group(luw, gN, D, False, st, 0)
    finally:
        capi.inject_fault(0)
        if saved is None: os.environ.pop("LUW_GROUP_TRANSPORT", None)
        else: os.environ["LUW_GROUP_TRANSPORT"] = saved
        capi.reload_tuning()
    g = run_group(luw, gN, D, False, st, 0)
    assert g.direct_peer_stores()
    g.run(0); g.run(5); g.read_from_device()
    o = oracle.OracleLBM(*gN, 0.01)
    o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
    o.run(5)
    assert np.array_equal(g.u, o.u) and np.array_equal(g.rho, o.rho)
    g.close()
