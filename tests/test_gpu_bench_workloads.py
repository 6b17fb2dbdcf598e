"""The benchmark workloads THEMSELVES (bench.fill_channel: TYPE_E inflow shell, log-law profile, solid ground, building
array, LES at tau ~ 0.5) at full size through the product path, bit for bit against the CPU oracle: BASELINE configs[1]
(512^3 FP32), configs[2] (1024x1024x256 with the building array, FP32 and FP16C/pair kernel), and one rank of the 8-GPU tile
of configs[3] / configs[4] in its real local shape (2048x258x258 of n_gpu=[1,4,2] and 514x514x512 of the deck's literal
[4,2,1]; FP16C + Coriolis for configs[4]) with nudging + sponge, stepped through the production schedule and the real RCCL
transport.  The oracle moves ~75 M cells/s on the test box's 16 cores, so a few steps of each take seconds.  GPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def planes_equal(a, b, planes, fp16c=False):
    """plane by plane (no lattice-sized temporaries beyond one plane); FP16C codes 0x0000 / 0x8000 are both zero"""
    a = a.reshape(planes, -1); b = b.reshape(planes, -1)
    for i in range(planes):
        if fp16c:
            x, y = a[i], b[i]
            if not np.array_equal(np.where(x == 0x8000, 0, x), np.where(y == 0x8000, 0, y)):
                return False
        elif not np.array_equal(a[i], b[i]):
            return False
    return True


def run_workload_vs_oracle(luw, size, fp16c, buildings, chunks=(3, 1), coriolis=False):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from bench import fill_channel, coriolis_omega, NU
    from oracle import oracle
    Nx, Ny, Nz = size
    g = luw.LBM(Nx, Ny, Nz, NU, fp16c=fp16c)                    # exactly what bench.run_single creates (automatic kernel choice)
    try:
        fill_channel(g.flags.data, g.u.data, g.rho.data, Nx, Ny, Nz, buildings=buildings)
        o = oracle.OracleLBM(Nx, Ny, Nz, NU, fp16c=fp16c)
        o.flags[:] = g.flags.data; o.u[:] = g.u.data; o.rho[:] = g.rho.data
        if coriolis:
            g.set_coriolis(*coriolis_omega()); o.set_coriolis(*coriolis_omega())
        g.run(0); o.initialize()
        for k in chunks:                                        # 3 + 1 steps: both time parities, and a run() call that starts on an odd t
            g.run(k); o.run(k)
            g.u.read_from_device(); g.rho.read_from_device()
            assert np.array_equal(g.rho.data, o.rho), "rho differs at t=%d" % o.t
            assert planes_equal(g.u.data, o.u, 3), "u differs at t=%d" % o.t
        fi = g.download_fi()
        assert planes_equal(fi, o.fi, 19, fp16c), "DDFs differ at t=%d" % o.t
        assert np.isfinite(g.u.data).all()
        if buildings:                                           # not a trivial state: the buildings deflect the flow upwards
            assert float(np.abs(g.u.data.reshape(3, -1)[2]).max()) > 0.0
    finally:
        g.close()


def test_c2_512cubed_fp32_vs_oracle(luw):
    """BASELINE configs[1]: the 512^3 empty channel of bench.py --workload c2, FP32 DDFs"""
    run_workload_vs_oracle(luw, (512, 512, 512), False, False)


@pytest.mark.parametrize("fp16c", [False, True])
def test_c3_building_cluster_vs_oracle(luw, fp16c):
    """BASELINE configs[2] = the driver's N = 1 bench line: 1024x1024x256 with the building array; FP32 (scalar kernel) and FP16C
    (pair kernel)"""
    run_workload_vs_oracle(luw, (1024, 1024, 256), fp16c, True)


def test_c3_fp16c_coriolis_vs_oracle(luw):
    """configs[4]'s physics on the largest single-GPU lattice: FP16C DDFs + Coriolis force (every cell on the forced path)"""
    run_workload_vs_oracle(luw, (1024, 1024, 256), True, True, chunks=(3,), coriolis=True)


def test_c3_shipped_configuration_thermal_fp16c_vs_oracle(luw):
    """what the reference's SHIPPED build runs on configs[2]'s lattice: FP16C DDFs AND the thermal D3Q7 lattice (scalar kernel with
    the thermal cell update), here with heat sources (TYPE_T presets on 3 % of the fluid cells): rho, u, T, all 19 + 7 planes"""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from bench import fill_channel, NU
    from helpers import thermal_state
    from oracle import oracle
    Nx, Ny, Nz = 1024, 1024, 256
    g = luw.LBM(Nx, Ny, Nz, NU, fp16c=True, alpha=2.1e-7)
    try:
        fill_channel(g.flags.data, g.u.data, g.rho.data, Nx, Ny, Nz, buildings=True)
        tflags, T = thermal_state(g.flags.data, (Nx, Ny, Nz))
        g.flags.data[:] = tflags; g.T.data[:] = T
        o = oracle.OracleLBM(Nx, Ny, Nz, NU, fp16c=True, alpha=2.1e-7)
        o.flags[:] = tflags; o.u[:] = g.u.data; o.rho[:] = g.rho.data; o.T[:] = T
        g.run(0); o.initialize()
        g.run(3); o.run(3)
        g.u.read_from_device(); g.rho.read_from_device(); g.T.read_from_device()
        assert np.array_equal(g.rho.data, o.rho) and planes_equal(g.u.data, o.u, 3)
        assert np.array_equal(g.T.data, o.T) and float(o.T.std()) > 1e-4
        assert planes_equal(g.download_fi(), o.fi, 19, True) and planes_equal(g.download_gi(), o.gi, 7, True)
    finally:
        g.close()


RANK_CASES = [
    # dtype, per-GPU block, n_gpu, rank, options
    ("f32", (2048, 256, 256), (1, 4, 2), 0, ("bld", "forcing")),            # configs[3], bench default cut; rank 0: ground, west/east/south faces
    # configs[3], the deck's literal n_gpu; last rank: east (downstream) + north faces, whole height
    ("f32", (512, 512, 512), (4, 2, 1), 7, ("bld", "forcing")),
    ("fp16c", (2048, 256, 256), (1, 4, 2), 7, ("bld", "forcing", "cor")),   # configs[4]; last rank: top sponge + north face
    ("fp16c", (512, 512, 512), (4, 2, 1), 0, ("bld", "forcing", "cor")),    # configs[4], literal n_gpu
    # the same production schedule with the faces written in place (PeerLoopbackTransport: the one-process host's peer stores, the rank its own neighbour;
    # x faces straight from the step kernels), exact and native arithmetic (native: within the tolerance of one FP16C code over 3 steps is not asked here --
    # the oracle comparison is for the exact kernels; the native run must equal ITS OWN RCCL-self run bit for bit: same kernels, another transport)
    ("f32", (384, 96, 64), (4, 2, 1), 3, ("bld", "forcing", "peer")),
    ("fp16c", (384, 96, 64), (4, 2, 1), 0, ("bld", "forcing", "cor", "peer")),
    ("fp16c", (640, 64, 64), (1, 4, 2), 5, ("bld", "forcing", "cor", "peer")),
    # the start-up schedule probe of `bench.py --gpus N` and switches between the two step schedules in the middle of a run (round 6): RCCL self copies and
    # faces written in place
    ("f32", (384, 96, 64), (4, 2, 1), 0, ("bld", "forcing", "switch")),
    ("fp16c", (384, 96, 64), (4, 2, 1), 5, ("bld", "forcing", "cor", "peer", "switch")),
]


@pytest.mark.parametrize("dt,block,D,rank,opts", RANK_CASES)
def test_rank_of_the_8gpu_tile_vs_oracle(dt, block, D, rank, opts):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + (os.getpid() + 7 * rank + len(opts)) % 150), RANK="0", WORLD_SIZE="1",
        LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "rank_shape_worker.py"), dt, *map(str, block), *map(str, D), str(rank), "3", *opts]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "DDFs equal True" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]
