"""A decomposed step pipelined ALONG z (opt-in: LUW_STEP_SCHEDULE=zchunks), for the one-process-per-GPU host (latticeurbanwind_amd.distributed) on domains
cut in x and / or y with z whole.

The default schedule (csrc/luw_step.hpp) splits a domain into a boundary shell -- 128-cell x slabs and one-cell y layers -- whose faces travel while the
interior is stepped.  The slabs are half the cells of a 512^3 rank and run 4-7 % under the whole-row rate (every row segment ends in a memory line it shares
with the neighbouring box, fetched twice).  Here a step is C launches of WHOLE rows instead, one per z range ("chunk"), one after the other on the compute
stream; the faces of a chunk -- its z range of the x faces, written by the step kernel itself, of the y faces and of the (x, y) edge lines -- travel while
the following chunks are stepped, and are needed by the neighbours only one step later.  No box of the step is narrower than the lattice.

Dependencies (chunk k touches lattice slots and face elements of the z layers of chunks k-1, k, k+1 only; z wraps periodically in the kernel, so the
first and the last chunk are neighbours too):
  step t+1, chunk k     needs the inserts I(k-1), I(k), I(k+1) of step t (y halo rows, edge lines; the x faces are read where they arrived, in the receive
                        set of step t, which I(m) follows on the communication stream)
  pack / send, chunk k  needs chunk k of this step
  insert I(m), step t   needs the arrival of chunk m's faces and the END of chunks m-1, m, m+1 of step t (they still read what step t-1 put there)
Every step starts one chunk further (order s, s+1, ..., s-1; next step s+1, ...): its first chunk then neighbours neither the chunk computed last nor its
faces still on the wire.  Two sets of x receive buffers alternate by step (a step reads one while the neighbours fill the other).  Same values as every
other schedule, bit for bit (tests/test_gpu_zchunks.py, schedule fuzzing included)."""
import os


def wanted():
    return os.environ.get("LUW_STEP_SCHEDULE", "") == "zchunks"


def chunk_count():
    return max(4, int(os.environ.get("LUW_ZCHUNKS", "4")))


def supported(sim, sample):
    """plain steps of a domain cut in x / y only, through the one-round exchange, with the library's kernels"""
    b, lay = sim.backend, sim.layout
    if not (sample is None and sim.pre_step is None and sim.one_phase and sim.overlap and hasattr(b, "face_range") and not getattr(b, "thermal", False)
            and 2 not in lay.split_axes() and len(lay.split_axes()) > 0 and lay.lN[2] >= 2 * chunk_count() and lay.lN[0] >= 6):
        return False
    if 0 not in lay.split_axes():
        return True
    # x cut: EVERY launch of the step has to write its x faces itself and read the incoming ones where they arrive (no pack / unpack kernel sees a
    # whole x face here) -- not every instantiation does (FP16C rows of an odd number of cells, the uniform-force pair kernel): ask the library
    Nx, Ny, Nz = lay.lN
    hy = 1 if 1 in lay.split_axes() else 0
    return bool(getattr(b, "x_insert_fused", False)) and all(b.lbm.launch_x_face_caps((1, Nx - 1, hy, Ny - hy, 0, Nz // chunk_count()), wf) == (True, True)
        for wf in (False, True))


def neighbours(m, C):
    """the chunks whose z layers chunk m's cells, faces and edge lines touch (z is periodic in the kernel)"""
    return {(m - 1) % C, m, (m + 1) % C}


def comm_plan(order, C):
    """what the communication stream does in a step whose chunks are computed in `order`: [("wait", k) | ("insert", m) | ("send", k)].  After the wait for
    chunk k: first the inserts that only waited for k to end (the next step starts on them -- they must not queue behind k's faces on the wire), then k's
    pack + send, then k's own insert if its neighbours have ended already.  Pure logic, checked by tests/test_zchunks_plan.py."""
    plan, computed, sent, inserted = [], set(), set(), set()
    for k in order:
        plan.append(("wait", k)); computed.add(k)
        for m in order:
            if m != k and m in sent and m not in inserted and neighbours(m, C) <= computed:
                plan.append(("insert", m)); inserted.add(m)
        plan.append(("send", k)); sent.add(k)
        if neighbours(k, C) <= computed:
            plan.append(("insert", k)); inserted.add(k)
    return plan


def run(sim, steps, timed=False):
    import torch
    b, lay = sim.backend, sim.layout
    comp, comm = b.compute, b.comm
    C = chunk_count()
    Nx, Ny, Nz = lay.lN
    axes = lay.split_axes()
    zr = [(Nz * k // C, Nz * (k + 1) // C) for k in range(C)]
    hx, hy = (1 if 0 in axes else 0), (1 if 1 in axes else 0)
    boxes = [(hx, Nx - hx, hy, Ny - hy, z0, z1) for z0, z1 in zr]
    if 0 in axes: b.two_x_receive_sets()
    edges = sorted(b.ebuf)
    state = sim.__dict__.setdefault("_zchunk_state", {"start": 0, "inserted": {}})
    from .layout import C19
    ev_k, ev_c = [], []
    # whatever the previous call or the initialisation left on either stream (the unpack of the first exchange runs on the communication stream) comes first
    comp.wait_stream(comm); comm.wait_stream(comp)
    for i in range(steps):
        wf = (i + 1 == steps) or sim._every_step()
        order = [(state["start"] + j) % C for j in range(C)]
        done = {}
        if timed:
            e0 = torch.cuda.Event(enable_timing=True); e0.record(comp)
        for k in order:                                          # 1. the chunks, whole rows, on the compute stream
            for m in sorted(neighbours(k, C)):
                if m in state["inserted"]: comp.wait_event(state["inserted"][m])
            b.stream_collide(boxes[k], wf, comp)
            done[k] = torch.cuda.Event(); done[k].record(comp)
        if timed:
            e1 = torch.cuda.Event(enable_timing=True); e1.record(comp); ev_k.append((e0, e1))
            x0 = torch.cuda.Event(enable_timing=True); x0.record(comm)
        inserted = {}
        inputs_pending = [False]

        def insert(m):
            if 0 in axes and not inputs_pending[0]:
                # (from here on the edges across the x cut join the faces that have just arrived: csrc/luw_kernels_aux.hpp k_edges)
                b.x_inputs_pending(comm); inputs_pending[0] = True
            z0, z1 = zr[m]
            if 1 in axes: b.face_range(1, comm, z0 * Nx, (z1 - z0) * Nx, insert=True)
            if edges: b.edges_range(comm, z0, z1 - z0, insert=True)
            inserted[m] = torch.cuda.Event(); inserted[m].record(comm)

        def send(k):
            z0, z1 = zr[k]
            msgs = []
            for a in axes:
                sp, sm, rp, rm = b.buf[a][:4]
                A, first, count = (Ny * Nz, z0 * Ny, (z1 - z0) * Ny) if a == 0 else (Nx * Nz, z0 * Nx, (z1 - z0) * Nx)
                if a == 1: b.face_range(1, comm, first, count, insert=False)
                unit = tuple(1 if q == a else 0 for q in range(3))
                for p in range(5):
                    sl = slice(p * A + first, p * A + first + count)
                    msgs += [(sp[sl], rm[sl], unit), (sm[sl], rp[sl], tuple(-v for v in unit))]
            if edges:
                b.edges_range(comm, z0, z1 - z0, insert=False)
                msgs += [(b.ebuf[e][0][z0:z1], b.ebuf[e][1][z0:z1], C19[7 + e]) for e in edges]
            with torch.cuda.stream(comm):
                sim.transport.exchange_all(msgs)

        for what, k in comm_plan(order, C):                      # 2. their faces, on the communication stream
            if what == "wait": comm.wait_event(done[k])
            elif what == "insert": insert(k)
            else: send(k)
        assert len(inserted) == C
        if 0 in axes: b.next_x_buffers()
        if timed:
            x1 = torch.cuda.Event(enable_timing=True); x1.record(comm); ev_c.append((x0, x1))
        b.increment_time_step(1)
        state["inserted"] = inserted
        state["start"] = (state["start"] + 1) % C
    sim._join()
    state["inserted"] = {}                                       # (everything has arrived: the next call waits for nothing)
    if timed and steps:
        mean = lambda pairs: sum(a.elapsed_time(c) for a, c in pairs) / len(pairs)
        return {"kernel_ms": mean(ev_k), "shell_ms": None, "exchange_ms": mean(ev_c)}
    return None
