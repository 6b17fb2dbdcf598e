"""The whole multi-domain machinery on ONE GPU: 2..8 HipDomain objects (real pack/unpack kernels, shell/interior split
on two streams, box launches of the vector kernel) driven in lock-step must equal the single-domain HIP run and the
oracle bit for bit.  GPU only (what cannot run here -- RCCL between processes -- is covered by the gloo test of the
same driver code)."""
import os

import numpy as np
import pytest

from helpers import synthetic_state

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gN,D", [((24, 20, 16), (2, 1, 1)), ((24, 20, 16), (2, 2, 2)), ((32, 24, 12), (4, 2, 1)), ((26, 18, 16), (1, 3, 2))])
@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("fp16c", [False, True])
def test_local_group_equals_single_domain(luw, gN, D, overlap, fp16c):
    from latticeurbanwind_amd.distributed import LocalGroup, HipDomain
    from oracle import oracle
    flags, u, rho = synthetic_state(*gN, seed=31, shell=None)
    steps = 5
    grp = LocalGroup(gN, D, 0.01, lambda lay: HipDomain(lay, 0.01, fp16c=fp16c), overlap=overlap)
    assert all(s.overlap == overlap for s in grp.sims)
    for s in grp.sims:
        s.set_fields_from_global(flags, u, rho)
    grp.run(steps)
    gu, grho = grp.gather_u_rho()
    o = oracle.OracleLBM(*gN, 0.01, fp16c=fp16c)
    o.flags[:] = flags; o.u[:] = u; o.rho[:] = rho
    o.run(steps)
    assert np.array_equal(gu, o.u) and np.array_equal(grho, o.rho)
    for s in grp.sims:
        s.backend.lbm.close()


@pytest.mark.parametrize("gN,D,overlap", [((24, 20, 16), (1, 2, 2), True), ((26, 18, 16), (1, 3, 2), False), ((64, 12, 8), (1, 2, 1), True),
                                          ((24, 20, 16), (2, 1, 1), False), ((32, 24, 12), (4, 2, 1), False), ((27, 18, 16), (1, 3, 2), True)])
def test_local_group_pair_kernel(luw, gN, D, overlap):
    """the FP16C pair kernel (two cells per lane; the automatic choice for wide FP16C rows) in decomposed runs: halo cells pass
    through, shell and interior boxes tile the domain; x whole (also with an odd row length) and x split (pairs then start at the
    first owned cell, x = 1)"""
    from latticeurbanwind_amd import capi
    from latticeurbanwind_amd.distributed import LocalGroup, HipDomain
    from oracle import oracle
    flags, u, rho = synthetic_state(*gN, seed=33, shell=None)
    steps = 6
    grp = LocalGroup(gN, D, 0.01, lambda lay: HipDomain(lay, 0.01, fp16c=True, kernel=capi.KERNEL_PAIR), overlap=overlap)
    for s in grp.sims:
        s.set_fields_from_global(flags, u, rho)
    grp.run(steps)
    gu, grho = grp.gather_u_rho()
    o = oracle.OracleLBM(*gN, 0.01, fp16c=True)
    o.flags[:] = flags; o.u[:] = u; o.rho[:] = rho
    o.run(steps)
    assert np.array_equal(gu, o.u) and np.array_equal(grho, o.rho)
    for s in grp.sims:
        s.backend.lbm.close()


@pytest.mark.parametrize("D,overlap,fp16c", [((2, 1, 1), False, False), ((1, 2, 2), True, False), ((2, 2, 2), True, True)])
def test_local_group_thermal_lattice(luw, D, overlap, fp16c):
    """thermal D3Q7 lattice across domains on the GPU (pack / unpack of the single gi population per face cell): T and u of
    2-8 HIP domains equal the single-domain oracle"""
    from latticeurbanwind_amd.distributed import LocalGroup, HipDomain
    from oracle import oracle
    from helpers import thermal_state
    gN = (24, 20, 16)
    flags, u, rho = synthetic_state(*gN, seed=31, shell=None)
    tflags, T = thermal_state(flags, gN)
    grp = LocalGroup(gN, D, 0.01, lambda lay: HipDomain(lay, 0.01, fp16c=fp16c, alpha=0.004), overlap=overlap)
    for s in grp.sims:
        s.set_fields_from_global(tflags, u, rho, T)
    grp.run(6)
    gu, _ = grp.gather_u_rho()
    gT = np.zeros((1, gN[2], gN[1], gN[0]), np.float32)
    for s in grp.sims:
        tb, off = s.interior_to_global(s.backend.download_T(), 1)
        gT[:, off[2]:off[2] + tb.shape[1], off[1]:off[1] + tb.shape[2], off[0]:off[0] + tb.shape[3]] = tb
    o = oracle.OracleLBM(*gN, 0.01, fp16c=fp16c, alpha=0.004)
    o.flags[:] = tflags; o.u[:] = u; o.rho[:] = rho; o.T[:] = T
    o.run(6)
    assert np.array_equal(gu, o.u) and np.array_equal(gT.ravel(), o.T)
    for s in grp.sims:
        s.backend.lbm.close()


def test_box_launches_tile_the_domain(luw):
    # any partition of the lattice into boxes gives the same result as one whole-domain launch
    from latticeurbanwind_amd import capi
    Nx, Ny, Nz = 45, 14, 9
    st = synthetic_state(Nx, Ny, Nz, seed=8, shell="luw")
    res = {}
    for name, kern, boxes in (("whole_s", capi.KERNEL_SCALAR, [(0, Nx, 0, Ny, 0, Nz)]),
                              ("split_a", capi.KERNEL_SCALAR, [(0, 1, 0, Ny, 0, Nz), (1, 7, 0, Ny, 0, Nz), (7, 30, 0, 5, 0, Nz), (7, 30, 5, Ny, 0, 4),
                                  (7, 30, 5, Ny, 4, Nz), (30, 44, 0, Ny, 0, Nz), (44, 45, 0, Ny, 0, Nz)]),
                              ("split_s", capi.KERNEL_SCALAR, [(0, 20, 0, Ny, 0, Nz), (20, Nx, 0, 6, 0, Nz), (20, Nx, 6, Ny, 0, Nz)])):
        g = luw.LBM(Nx, Ny, Nz, 1e-3, kernel=kern)
        g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
        g.run(0)
        for step in range(4):
            for b in boxes:
                g.enqueue_stream_collide(b, True)
            g.finish(); g.increment_time_step(1)
        g.u.read_from_device(); g.rho.read_from_device()
        res[name] = (g.u.data.copy(), g.rho.data.copy(), g.download_fi())
        g.close()
    for k in ("split_a", "split_s"):
        for a, b in zip(res["whole_s"], res[k]):
            assert np.array_equal(a, b), k


@pytest.mark.parametrize("dt", ["f32", "fp16c"])
def test_rccl_transport_through_self_send_recv(dt):
    """the production transport (RCCL batch_isend_irecv on the communication stream; FP16C codes as float16) on the test box's
    single GPU: a one-rank world whose neighbours are the rank itself -- tools/check_nccl_self.py asserts that the fields equal
    those of the in-process loopback run bit for bit"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200 + (1 if dt == "f32" else 0)), RANK="0", WORLD_SIZE="1",
        LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_nccl_self.py"), dt], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and "fields identical: True" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("fp16c,native,Nx,thermal", [(False, False, 40, False), (True, False, 322, False), (True, True, 322, False), (True, False, 322, True),
    (True, True, 322, True), (False, False, 40, True)])
@pytest.mark.parametrize("forces", ["none", "zones+coriolis"])
def test_x_faces_written_by_the_step_kernels_equal_the_extract_kernel(luw, fp16c, native, Nx, forces, thermal):
    """luw_set_x_face_buffers: the step kernels that hold the first / last owned x column put that column's five outgoing populations into the face buffers
    themselves.  Against the pack kernel reading the lattice behind the same step: every element whose source cell is an owned cell -- collided or not
    (solids on the border columns forward what their slots hold: bounce-back across a cut) -- must be the same value; with the buffers set, the extract
    call on them launches nothing; sampled steps (no such instantiation) fall back to the pack kernel by themselves."""
    import torch
    from helpers import TYPE_S
    Ny, Nz = 12, 10
    st = synthetic_state(Nx, Ny, Nz, seed=17, shell=None)
    flags = st[0].reshape(Nz, Ny, Nx).copy()
    # solids on both border columns
    flags[2:5, 3:7, 1] = TYPE_S; flags[4:8, 2:5, Nx - 2] = TYPE_S; flags[6, 6, 1:3] = TYPE_S; flags[3, 8, Nx - 3:Nx - 1] = TYPE_S
    kw = dict(buffer_nudging=dict(n_cells=3, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1),
        top_sponge=dict(n_cells=3, inv_tau=0.02)) if "zones" in forces \
        else {}
    if thermal: kw = dict(kw, alpha=0.004)        # the pair kernel with the thermal lattice (the shipped build's configuration) writes the D3Q19 faces too
    g = luw.LBM(Nx, Ny, Nz, 0.01, fp16c=fp16c, D=(2, 2, 1), O=(-1, -1, 0), native_arith=native, **kw)
    if "coriolis" in forces: g.set_coriolis(0.0, 3e-5, 4e-5)
    g.flags.data[:] = flags.ravel(); g.u.data[:] = st[1]; g.rho.data[:] = st[2]
    g.run(0)
    tdt = torch.int16 if fp16c else torch.float32
    A = g.area(0)
    fused = [torch.full((5 * A,), 77, dtype=tdt, device="cuda") for _ in range(2)]
    plain = [torch.zeros(5 * A, dtype=tdt, device="cuda") for _ in range(2)]
    g.set_x_face_buffers(fused[0].data_ptr(), fused[1].data_ptr())
    owned = np.zeros((Nz, Ny), bool); owned[:, 1:Ny - 1] = True                     # source rows that a launch covers (y is split: rows 0 and Ny-1 are halo)
    sel = np.tile(owned.ravel(), 5)
    x_slab = 128 if fp16c else 16
    boxes = [(1, 1 + x_slab, 1, Ny - 1, 0, Nz), (1 + x_slab, Nx - 1 - x_slab, 1, Ny - 1, 0, Nz), (Nx - 1 - x_slab, Nx - 1, 1, Ny - 1, 0, Nz)]
    for step in range(4):
        for b in boxes:
            g.enqueue_stream_collide(b, False)
        g.enqueue_extract_fi(0, plain[0].data_ptr(), plain[1].data_ptr()); g.finish()          # other buffers: the pack kernel runs
        for f, p_ in zip(fused, plain):
            assert np.array_equal(f.cpu().numpy()[sel], p_.cpu().numpy()[sel]), "step %d" % step
            assert (f.cpu().numpy()[~sel] == 77).all()                                        # rim rows are nobody's: untouched
        before = [f.clone() for f in fused]
        g.enqueue_extract_fi(0, fused[0].data_ptr(), fused[1].data_ptr()); g.finish()          # covered by the launches: nothing to do
        assert all(torch.equal(a, b) for a, b in zip(before, fused))
        g.increment_time_step(1)
    # disjoint boxes that cut the columns cover them together (the step without x slabs: y / z layers and interior); rim rows stay nobody's
    g.enqueue_stream_collide((1, Nx - 1, 1, Ny // 2, 0, Nz), False); g.enqueue_stream_collide((1, Nx - 1, Ny // 2, Ny - 1, 0, Nz), False)
    g.finish(); before = [f.clone() for f in fused]
    g.enqueue_extract_fi(0, fused[0].data_ptr(), fused[1].data_ptr()); g.enqueue_extract_fi(0, plain[0].data_ptr(), plain[1].data_ptr()); g.finish()
    assert all(torch.equal(a, b) for a, b in zip(before, fused))
    for f, p_ in zip(fused, plain): assert np.array_equal(f.cpu().numpy()[sel], p_.cpu().numpy()[sel])
    g.increment_time_step(1)
    # part of the columns, or a box launched twice, is no cover: the extract call does the work itself (rim rows included)
    for twice in (False, True):
        g.enqueue_stream_collide((1, Nx - 1, 1, Ny // 2, 0, Nz), False)
        if twice: g.enqueue_stream_collide((1, Nx - 1, 1, Ny // 2, 0, Nz), False); g.enqueue_stream_collide((1, Nx - 1, Ny // 2, Ny - 1, 0, Nz), False)
        g.enqueue_extract_fi(0, fused[0].data_ptr(), fused[1].data_ptr()); g.enqueue_extract_fi(0, plain[0].data_ptr(), plain[1].data_ptr()); g.finish()
        assert all(torch.equal(a, b) for a, b in zip(plain, fused)), twice
        g.increment_time_step(1)
    g.close()


@pytest.mark.parametrize("fp16c", [False, True])
@pytest.mark.parametrize("D", [(2, 2, 2), (4, 2, 1), (1, 2, 2)])
def test_edge_kernels_move_the_lines_the_two_hop_route_moves(luw, fp16c, D):
    """luw_enqueue_extract_edges / luw_enqueue_insert_edges (csrc/luw_kernels_aux.hpp k_edges) against the slot algebra restated in numpy
    (tests/oracle_domain.py _edge_line, which tests/test_distributed_gloo.py holds to the undivided run): for both parities of t, every edge this domain
    has reads exactly that line of the lattice, and writes exactly that line and nothing else."""
    import torch
    from latticeurbanwind_amd.distributed import DomainLayout
    from oracle_domain import OracleDomain
    gN = tuple(d * n for d, n in zip(D, (300, 7, 5)))        # a line longer than one block of the kernel
    lay = DomainLayout(gN, D, 0)
    od = OracleDomain(lay, 0.01, fp16c=fp16c)
    g = luw.LBM(*lay.lN, 0.01, fp16c=fp16c, D=D, O=lay.O)
    rng = np.random.default_rng(4)
    n = 19 * od.o.N
    fi = rng.integers(1, 60000, n).astype(np.uint16) if fp16c else rng.random(n, dtype=np.float32)
    tdt = torch.int16 if fp16c else torch.float32
    edges = lay.edges()
    assert len(edges) == {3: 12, 2: 4}[len(lay.split_axes())] and all(g.edge_length(e) == (lay.edge_length(e) if e in edges else 0) for e in range(12))
    for t in (0, 1):
        od.o.t = t
        g.upload_fi(fi)
        out = {e: torch.full((g.edge_length(e),), 7, dtype=tdt, device="cuda") for e in edges}
        g.enqueue_edges([out[e].data_ptr() if e in out else 0 for e in range(12)], insert=False); g.finish()
        for e in edges:
            assert np.array_equal(out[e].cpu().numpy().view(fi.dtype), fi[od._edge_line(e, True)]), (t, e)
        msg = {e: (rng.integers(1, 60000, g.edge_length(e)).astype(np.uint16) if fp16c else rng.random(g.edge_length(e), dtype=np.float32)) for e in edges}
        dev = {e: torch.from_numpy(m.view(np.int16) if fp16c else m).cuda() for e, m in msg.items()}
        g.enqueue_edges([dev[e].data_ptr() if e in dev else 0 for e in range(12)], insert=True); g.finish()
        want = fi.copy()
        for e in edges:
            want[od._edge_line(e, False)] = msg[e]
        assert np.array_equal(np.asarray(g.download_fi()), want), t
        g.increment_time_step(1)
    g.enqueue_edges([0] * 12, insert=False); g.finish()     # null entries: those edges are not moved by the call
    g.close()


@pytest.mark.parametrize("fp16c,native,Nx", [(False, False, 42), (True, False, 322), (True, True, 322)])
@pytest.mark.parametrize("forces", ["none", "coriolis", "zones+coriolis"])
def test_x_faces_read_from_the_receive_buffers_equal_the_insert_kernel(luw, monkeypatch, fp16c, native, Nx, forces):
    """luw_set_x_face_inputs: the step kernels of an x-split rank take the x faces of the last exchange from the receive buffers (edges across the x cut in
    their rims) instead of from the lattice behind the unpack kernel.  The production host on one rank that is its own neighbour, a few plain steps, then
    sampled ones (kernels that cannot read the buffers: the library runs the unpack kernel by itself), solids on both border columns; the FP16C domain is
    322 cells wide, so that its second x slab (64 cells) takes the one-cell kernel, which cannot read the buffers either: that side alone goes through the
    unpack kernel.  Same DDFs, rho, u as with LUW_X_INSERT_FUSED=0, which the rank-shape tests hold to the oracle."""
    from latticeurbanwind_amd.distributed import DomainDecomposedLBM, DomainLayout, PeerLoopbackTransport
    from helpers import TYPE_S, TYPE_E
    D, own = (2, 2, 1), (Nx - 2, 10, 9)
    gN = tuple(o * d for o, d in zip(own, D))
    monkeypatch.setenv("LUW_X_SHELL", "128" if fp16c else "16")
    kw = dict(buffer_nudging=dict(n_cells=3, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1), top_sponge=dict(n_cells=3, inv_tau=0.02)) \
        if "zones" in forces else {}
    from latticeurbanwind_amd import capi
    if not fp16c and forces == "coriolis":      # the FP32 kernel's row addressing form (what lattices with planes beyond 4 GiB take) in this one case
        monkeypatch.setenv("LUW_TEST_AIDS", "addr_row")
    capi.reload_tuning()
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("LUW_X_INSERT_FUSED", fused)
        lay = DomainLayout(gN, D, 0)
        sim = DomainDecomposedLBM(gN, D, 0.01, rank=0, transport=PeerLoopbackTransport(lay), fp16c=fp16c, device=0, native_arith=native, **kw)
        assert sim.overlap and sim.one_phase and sim.backend.x_insert_fused == (fused == "1")
        lx, ly, lz = lay.lN
        st = synthetic_state(lx, ly, lz, seed=23, shell=None)
        flags = st[0].reshape(lz, ly, lx).copy()
        if "zones" in forces:      # the global faces this rank owns are inputs (reference cells of the zones: otherwise a step reads fields it is rewriting)
            flags[:, :, 1] = TYPE_E; flags[:, 1, :] = TYPE_E; flags[lz - 1, :, :] = TYPE_E
        flags[2:5, 3:7, 1] = TYPE_S; flags[4:8, 2:5, lx - 2] = TYPE_S; flags[6, 6, 1:3] = TYPE_S; flags[3, 8, lx - 3:lx - 1] = TYPE_S
        sim.set_fields(flags.ravel(), st[1], st[2])
        if "coriolis" in forces: sim.backend.set_coriolis(0.0, 3e-5, 4e-5)
        sim.run(5)
        assert not sim.backend.lbm.fields_every_step()      # (every reference cell is an input)
        sim.backend.stats_reset()
        sim.run(4, sample=(2, 2))
        sim.run(3)
        u, rho = sim.fields()
        # DDF slots located at owned cells that are collided (a solid border cell's slots, and the halo column's, are storage nobody reads: the insert kernel
        # fills them every step, the kernels that read the buffers forward from there)
        live = np.zeros((lz, ly, lx), bool); live[:, 1:ly - 1, 1:lx - 1] = True
        live &= flags != TYPE_S
        fi = np.asarray(sim.backend.lbm.download_fi()).reshape(19, lz, ly, lx)[:, live]
        res[fused] = (u.copy(), rho.copy(), fi.copy())
        sim.backend.close()
    monkeypatch.delenv("LUW_TEST_AIDS", raising=False)
    capi.reload_tuning()
    for a, b in zip(res["1"], res["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("fp16c,Nx", [(False, 42), (True, 322)])
@pytest.mark.parametrize("thermal", [False, True])
@pytest.mark.parametrize("D", [(2, 2, 1), (2, 1, 2)])
def test_one_phase_exchange_equals_the_three_phase_route_on_the_device(luw, monkeypatch, fp16c, Nx, thermal, D):
    """the production host on one rank that is its own neighbour: everything in one batch with edge messages and the x faces left in their buffers (default)
    against the reference's three phases with rims and unpack kernels (LUW_EXCHANGE=sequential) -- with the thermal lattice too, whose kernels neither write
    nor read x faces themselves (pack kernels; the pending x faces go through the unpack kernel in front of the next step): same rho, u, T, DDFs."""
    from latticeurbanwind_amd.distributed import DomainDecomposedLBM, DomainLayout, PeerLoopbackTransport
    from helpers import thermal_state
    own = (Nx - 2, 10, 9) if D[1] > 1 else (Nx - 2, 9, 10)
    gN = tuple(o * d for o, d in zip(own, D))
    monkeypatch.setenv("LUW_X_SHELL", "128" if fp16c else "16")
    res = {}
    for exchange in ("batch", "sequential"):
        monkeypatch.setenv("LUW_EXCHANGE", exchange)
        lay = DomainLayout(gN, D, 0)
        sim = DomainDecomposedLBM(gN, D, 0.01, rank=0, transport=PeerLoopbackTransport(lay), fp16c=fp16c, device=0, **(dict(alpha=0.004) if thermal else {}))
        assert sim.overlap and sim.one_phase == (exchange == "batch")
        lx, ly, lz = lay.lN
        st = synthetic_state(lx, ly, lz, seed=31, shell=None)
        if thermal:
            tflags, T = thermal_state(st[0], (lx, ly, lz))
            sim.set_fields(tflags, st[1], st[2], T)
        else:
            sim.set_fields(st[0], st[1], st[2])
        sim.run(7)
        u, rho = sim.fields()
        out = [u.copy(), rho.copy(), np.asarray(sim.backend.lbm.download_fi()).copy()]
        if thermal: out += [sim.backend.download_T().copy(), np.asarray(sim.backend.lbm.download_gi()).copy()]
        res[exchange] = out
        sim.backend.close()
    assert len(res["batch"]) == (5 if thermal else 3)
    for a, b in zip(res["batch"], res["sequential"]):
        assert np.array_equal(a, b)
