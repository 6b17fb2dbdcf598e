"""Domain-decomposed runs: one process per GPU, one `LBM` domain per process, one-cell halos exchanged with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).

What is reproduced from the reference (FX/lbm.cpp:1057-1073,1242-1290,1907-1935; FX/kernel.cpp:2188-2270):
  * block decomposition Dx x Dy x Dz (deck key n_gpu), domain id d = x + (y + z*Dy)*Dx, local extents N/D + 2 on
    split axes, offsets O = coord*N/D - 1, periodic neighbour (x+1)%Dx;
  * per step: stream_collide on every non-halo cell, then for axis x, y, z in this order: extract the 5 outgoing DDFs
    of both faces (face areas include the halo rims, so edge/corner data travel in up to 3 hops), swap with the two
    neighbours, insert; then t++.  LBM::initialize does the same exchange once with an odd t.
What is ours: all faces travel in ONE batch -- the populations that cross two cuts at once (one per diagonal direction in D3Q19, along the line where two
faces meet) go straight to the diagonal neighbour as 12 short edge messages instead of riding the rims of a second and third hop (csrc/luw_kernels_aux.hpp
k_edges; LUW_EXCHANGE=sequential restores the reference's three phases); the exchange never touches the host (the reference stages every face through
PCIe and swaps host pointers); the boundary shell (cells next to a halo) is computed first on a communication stream, its faces are packed and sent while
the interior is computed on the compute stream, and the next step starts when both are done.

Results are identical to a single-domain run of the global lattice (tests/test_distributed_gloo.py,
tests/test_gpu_halo.py).
"""
import numpy as np

from .hip_domain import HipDomain   # noqa: F401
from .layout import C19, DomainLayout, choose_decomposition, tile_lattice   # noqa: F401
from .transports import HostStagedTransport, PeerLoopbackTransport, SelfExchangeTransport, TorchDistTransport, init_rccl_process_group   # noqa: F401


class DomainDecomposedLBM:
    """The multi-domain `LBM` of the reference (FX/lbm.cpp:1057-1112,1221-1312) for THIS rank's domain."""

    def __init__(self, global_N, D, nu, rank=None, backend=None, transport=None, overlap=None, **backend_kw):
        if rank is None:
            import torch.distributed as dist
            rank = dist.get_rank()
        import os
        # x slabs of the boundary shell: 128 cells for both DDF formats (FP16C: a full wave of the pair kernel, 2 cells per lane, which narrower slabs
        # would leave to the one-cell kernel; FP32: no different from 64 within the run-to-run spread, profiles/r03_ab_rank_shape_xshell.txt); LUW_X_SHELL
        # overrides: A/B aid
        x_shell = int(os.environ.get("LUW_X_SHELL", "0")) or DomainLayout.X_SHELL
        self.layout = DomainLayout(global_N, D, rank, x_shell=x_shell)
        self.wire = {}           # standalone face-exchange rates per split axis (TorchDistTransport.warm_up with LUW_MEASURE_WIRE=<repetitions>)
        self.lNx, self.lNy, self.lNz = self.layout.lN
        self.global_offset = self.layout.O
        if transport is None:
            import torch.distributed as dist
            staged = (backend is None or isinstance(backend, HipDomain)) and dist.is_initialized() and dist.get_backend() == "gloo"
            transport = HostStagedTransport(self.layout) if staged else TorchDistTransport(self.layout)
            if backend is None and not staged and dist.is_initialized() and dist.get_backend() == "nccl":
                import torch      # connections to the neighbours first, the lattice second (see TorchDistTransport.warm_up)
                self.wire = transport.warm_up(torch.device("cuda", backend_kw.get("device", 0)), torch.float16 if backend_kw.get("fp16c") else torch.float32,
                                              measure=int(os.environ.get("LUW_MEASURE_WIRE", "0")))
        self.transport = transport
        self.backend = backend if backend is not None else HipDomain(self.layout, nu, alias_faces=getattr(transport, "alias_faces", False), **backend_kw)
        if overlap is None:
            # Shell/interior overlap.  x kept whole (whole-row shells): always.  x split: the 64-cell x slabs cost the step 4 %
            # (512^3 rank of n_gpu=[4,2,1] with pipelined steps: 3.94-3.98 ms against 3.78-3.89 for whole box + exchange when the
            # exchange is a device-local copy), which any real wire time (>= 0.3 ms for its four faces) outweighs; LUW_X_OVERLAP=0
            # restores whole box + exchange there.
            import os
            overlap = self.layout.D[0] == 1 or os.environ.get("LUW_X_OVERLAP", "1") != "0"
        self.overlap = bool(overlap) and self.layout.can_overlap() and hasattr(self.backend, "comm")
        if hasattr(self.backend, "configure_step"):
            self.backend.configure_step(self.overlap)            # the library's step context follows the choice (LUW_X_OVERLAP=0, tests)
        import os
        # all faces + the 12 edge populations in one batch (default), or the reference's three phases with rims (LUW_EXCHANGE=sequential; also what a
        # backend or transport without the edge calls gets)
        self.batch_wanted = os.environ.get("LUW_EXCHANGE", "batch") != "sequential"
        self.pipeline = os.environ.get("LUW_PIPELINE", "1") != "0"     # A/B switch: 0 = join both streams after every step
        self.initialized = False
        self.pre_step = None     # callable(stream) enqueued before every step's kernels (von-Karman inlet update)

    def set_schedule(self, overlap):
        """between two run() calls: boundary shell first + exchange beside the interior (True), or the whole box as ONE launch followed by the exchange
        (False).  Same values either way; which is faster depends on what the exchange costs on the wire."""
        ov = bool(overlap) and self.layout.can_overlap() and hasattr(self.backend, "comm")
        if ov != self.overlap:
            self._join()
            self.overlap = ov
            if hasattr(self.backend, "configure_step"):
                self.backend.configure_step(ov)
        return self.overlap

    def choose_schedule(self, steps=20, reduce_max=None, margin=0.02):
        """Start-up probe, like luw_create's placement probe: `steps` REAL steps of this run under each schedule (the state simply advances), wall time
        of the slowest rank (reduce_max: a callable that takes [ms, ms] and returns the element-wise maximum over the ranks -- EVERY rank calls it exactly
        once, whatever happened locally, so no rank waits for another's collective), and the default (shell first) is kept unless the whole-box schedule
        is more than `margin` faster.  Returns what was measured and kept; None where there is nothing to choose."""
        if not (self.layout.can_overlap() and hasattr(self.backend, "comm") and hasattr(self.backend, "configure_step")):
            return None
        import time
        import torch
        if not self.initialized:
            self.initialize()
        default, ms = self.overlap, {}
        for ov in (True, False):
            try:
                self.set_schedule(ov); self.run(3); torch.cuda.synchronize()
                t0 = time.perf_counter(); self.run(steps); torch.cuda.synchronize()
                ms[ov] = (time.perf_counter() - t0) / steps * 1e3
            except Exception:                                   # a schedule that cannot run here is never chosen
                ms[ov] = float("inf")
        vec = [ms[True], ms[False]]
        if reduce_max is not None:
            vec = [float(v) for v in reduce_max(vec)]
        usable = all(v == v and v != float("inf") for v in vec)
        pick = default if not usable else (False if vec[1] < vec[0] * (1.0 - margin) else True)
        self.set_schedule(pick)
        return {"shell_first_ms": vec[0], "whole_box_ms": vec[1], "kept": "shell first, exchange beside the interior" if self.overlap
            else "whole box, then the exchange", "probe_steps": steps,
                "rule": "the default (shell first) unless the whole box is more than %d %% faster on the slowest rank" % int(margin * 100)}

    @property
    def one_phase(self):
        return self.batch_wanted and hasattr(self.backend, "extract_edges") and hasattr(self.transport, "exchange_all")

    def set_fields(self, flags, u, rho, T=None):
        if T is not None: self.backend.set_fields(flags, u, rho, T)
        else: self.backend.set_fields(flags, u, rho)

    def set_fields_from_global(self, gflags, gu, grho=None, gT=None):
        """cut this rank's box (incl. periodic halos) out of global (z,y,x) arrays; grho=None means rho = 1"""
        ix, iy, iz = self.layout.local_slices()
        sel = np.ix_(iz, iy, ix)
        gN = self.layout.gN
        f = gflags.reshape(gN[2], gN[1], gN[0])[sel]
        r = grho.reshape(gN[2], gN[1], gN[0])[sel] if grho is not None else np.ones(f.shape, np.float32)
        u = np.stack([gu.reshape(3, gN[2], gN[1], gN[0])[c][sel] for c in range(3)])
        if gT is not None:
            self.backend.set_fields(f.ravel(), u.ravel(), r.ravel(), np.ascontiguousarray(gT.reshape(gN[2], gN[1], gN[0])[sel]).ravel())
        else:
            self.backend.set_fields(f.ravel(), u.ravel(), r.ravel())

    # ---- FX/lbm.cpp:1907-1935 for the DDF field, device to device
    def communicate_fi(self, stream=None):
        b = self.backend
        if self.one_phase:
            return self._communicate_one_phase(stream)
        for axis in self.layout.split_axes():
            sp, sm = b.extract(axis, stream)
            rp, rm = b.recv_buffers(axis)
            self._exchange(axis, sp, sm, rp, rm, stream)
            b.insert(axis, stream)
        if getattr(b, "thermal", False):      # communicate_gi, FX/lbm.cpp:1283,1952-1954: after the DDFs, same axis order
            for axis in self.layout.split_axes():
                sp, sm = b.extract_g(axis, stream)
                rp, rm = b.recv_buffers_g(axis)
                self._exchange(axis, sp, sm, rp, rm, stream)
                b.insert_g(axis, stream)

    def _communicate_one_phase(self, stream):
        """every face of every split axis, the edge populations and the thermal faces in ONE batch.  Message types in a fixed order (see
        TorchDistTransport.exchange_all): per axis the + face, the - face; the edges by number; per axis the thermal + face, - face.  The rims of the faces
        (halo rows of the other axes) travel along unused: what the reference's second and third hop carried in them is what the edge messages deliver,
        inserted after the faces (csrc/luw_kernels_aux.hpp k_edges)."""
        b = self.backend
        axes = self.layout.split_axes()
        unit = lambda a, s: tuple(s if k == a else 0 for k in range(3))
        msgs = []
        for a in axes:
            sp, sm = b.extract(a, stream)
            rp, rm = b.recv_buffers(a)
            msgs += [(sp, rm, unit(a, +1)), (sm, rp, unit(a, -1))]     # my + face is the - halo of the domain above
        msgs += [(s, r, C19[7 + e]) for e, s, r in b.extract_edges(stream)]
        thermal = getattr(b, "thermal", False)
        if thermal:
            for a in axes:
                sp, sm = b.extract_g(a, stream)
                rp, rm = b.recv_buffers_g(a)
                msgs += [(sp, rm, unit(a, +1)), (sm, rp, unit(a, -1))]
        if stream is not None:
            import torch
            with torch.cuda.stream(stream):
                self.transport.exchange_all(msgs)
        else:
            self.transport.exchange_all(msgs)
        for a in axes:
            (b.insert_deferred if hasattr(b, "insert_deferred") else b.insert)(a, stream)
        b.insert_edges(stream)
        if thermal:
            for a in axes:
                b.insert_g(a, stream)

    def _exchange(self, axis, sp, sm, rp, rm, stream):
        if stream is not None:
            import torch
            with torch.cuda.stream(stream):
                self.transport.exchange(axis, sp, sm, rp, rm)
        else:
            self.transport.exchange(axis, sp, sm, rp, rm)

    def initialize(self):
        b = self.backend
        b.initialize()
        b.increment_time_step(1)          # "the communicate calls at initialization need an odd time step", FX/lbm.cpp:1242
        self.communicate_fi(getattr(b, "comm", None))
        self._join()
        b.reset_time_step()               # FX/lbm.cpp:1258
        self.initialized = True

    def _every_step(self):
        """rho,u written by every step: a nudging / sponge reference cell of this rank's faces is a fluid cell (luw_fields_every_step)"""
        lbm = getattr(self.backend, "lbm", None)
        return bool(lbm is not None and lbm.fields_every_step())

    def _join(self):
        b = self.backend
        if hasattr(b, "comm"):
            b.comm.synchronize(); b.compute.synchronize()

    def run(self, steps, timed=False, sample=None):
        """`steps` steps.  sample = (first, stride): step number `first` of this call (from 1) and every stride-th after it are
        statistics samples (the purge_avg window, FX/setup.cpp:4252-4268): the boxes of such a step carry the Welford update of
        their cells themselves (LUW_WF_SAMPLE) or, where the kernels cannot (thermal lattice), the step writes rho,u and the
        statistics kernel follows on the compute stream -- either way no host synchronisation, the overlap of exchange and
        interior continues through the window."""
        if not self.initialized:
            self.initialize()
        b = self.backend
        lay = self.layout
        if getattr(b, "step", None) is not None and self.overlap == b.step_overlaps() and self.pipeline and self.pre_step is None:
            return self._run_library_schedule(steps, timed, sample)
        ev, ev_comm = [], []
        stats_done = None
        pipelined = self.overlap and self.pipeline
        if self.overlap:
            import torch
        if pipelined:
            shell_done, interior_done = torch.cuda.Event(), torch.cuda.Event()
            self._join()
        for i in range(steps):
            sampled = sample is not None and i + 1 >= sample[0] and (i + 1 - sample[0]) % sample[1] == 0
            # fused: every box of the step updates the statistics of its own cells (each cell always on the same stream, so
            # samples of a cell stay ordered); otherwise the step writes rho,u and the statistics kernel follows
            fused = sampled and hasattr(b, "stats_begin_sample") and b.stats_begin_sample()
            kw = {"sample": True} if fused else {}
            if fused: sampled = False
            wf = (i + 1 == steps) or sampled or self._every_step()
            if self.overlap:
                comm, comp = b.comm, b.compute
                if wf and stats_done is not None:
                    comm.wait_event(stats_done); stats_done = None     # this step's shell rewrites the rho,u the last sample reads
                if pipelined:
                    # Who needs what (slot algebra of extract_one/insert_one, luw_kernels_aux.hpp, against load_f/store_f):
                    # interior cells lie two layers inside the halo, they touch slots of interior and shell cells only, and
                    # never the planes the pack/unpack kernels use there (pack reads, in a shell cell next to a -y face,
                    # planes 3/4 7/8 11/12 17/18, the interior writes 13/14 into it; mirrored on the other faces).  So
                    #   interior(t) needs interior(t-1) [stream order] and shell(t-1);
                    #   shell(t)    needs interior(t-1), shell(t-1) and the unpack of step t-1 [stream order].
                    # The interior of the next step therefore starts as soon as the previous one ends, while the exchange of
                    # the previous step may still be on the wire: no bubble between steps on the compute stream.
                    if i > 0:
                        comp.wait_event(shell_done)
                        comm.wait_event(interior_done)
                    if self.pre_step is not None:
                        self.pre_step(comp)
                        pre = torch.cuda.Event(); pre.record(comp); comm.wait_event(pre)
                else:
                    comp.wait_stream(comm)                             # step t needs all of step t-1
                    if self.pre_step is not None:
                        self.pre_step(comp)
                    comm.wait_stream(comp)
                if timed:
                    s0, s1, s2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                    s0.record(comm)
                for box in lay.shell_boxes():
                    b.stream_collide(box, wf, comm, **kw)              # boundary shell first ...
                if timed:
                    s1.record(comm)
                if pipelined:
                    shell_done = torch.cuda.Event(); shell_done.record(comm)
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(comp)
                b.stream_collide(lay.interior_box(), wf, comp, **kw)   # ... interior overlaps the halo traffic
                if timed:
                    e1.record(comp); ev.append((e0, e1))
                if pipelined:
                    interior_done = torch.cuda.Event(); interior_done.record(comp)
                self.communicate_fi(comm)
                if timed:
                    s2.record(comm); ev_comm.append((s0, s1, s2))
                if sampled:
                    if pipelined:
                        comp.wait_event(shell_done)                    # the sample reads rho,u of shell and interior cells
                    else:
                        comp.wait_stream(comm)
                    b.stats_enqueue(comp)
                    stats_done = torch.cuda.Event(); stats_done.record(comp)
            else:
                st = getattr(b, "compute", None)
                if self.pre_step is not None:
                    self.pre_step(st)
                if timed and st is not None:
                    import torch
                    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                    e0.record(st)
                b.stream_collide(lay.whole_box(), wf, st, **kw)
                if timed and st is not None:
                    e1.record(st)
                self.communicate_fi(st)
                if timed and st is not None:
                    e2.record(st); ev.append((e0, e1)); ev_comm.append((None, e1, e2))
                if sampled:
                    if hasattr(b, "stats_enqueue") and st is not None: b.stats_enqueue(st)
                    else: b.stats_accumulate()
            b.increment_time_step(1)
        self._join()
        if timed and ev:
            # means over the steps of this call, taken with events on the streams the work was enqueued on: the interior (or
            # whole-box) kernel; the boundary-shell launches; pack + exchange + unpack of all split axes (time on the
            # communication stream, which includes waiting for the neighbours)
            mean = lambda pairs: sum(a.elapsed_time(c) for a, c in pairs) / len(pairs)
            out = {"kernel_ms": mean(ev), "shell_ms": None, "exchange_ms": None}
            if ev_comm:
                out["exchange_ms"] = mean([(s1, s2) for _, s1, s2 in ev_comm])
                if ev_comm[0][0] is not None:
                    out["shell_ms"] = mean([(s0, s1) for s0, s1, _ in ev_comm])
            return out
        return None

    def _run_library_schedule(self, steps, timed, sample):
        """the production path on a GPU: per step ONE call launches this domain's kernels in the library's schedule (luw_domain_step_launch: boundary shell on
        the communication stream, interior on the compute stream, pipelined; the same code luw_group_* runs), then the exchange follows on the
        communication stream.  What stays here is what differs between the hosts: who the neighbours are and how the faces travel."""
        b = self.backend
        import torch
        comm = b.comm if self.overlap else b.compute
        ev_comm = []
        for i in range(steps):
            sampled = sample is not None and i + 1 >= sample[0] and (i + 1 - sample[0]) % sample[1] == 0
            fused = sampled and b.stats_begin_sample()
            separate = sampled and not fused
            wf = int((i + 1 == steps) or separate or self._every_step()) | (2 if fused else 0)
            b.step_launch(wf, timed)
            if timed:
                x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                x0.record(comm)
            self.communicate_fi(comm)
            if timed:
                x1.record(comm); ev_comm.append((x0, x1))
            if separate:
                b.step_separate_stats()
            b.increment_time_step(1)
        self._join()
        if timed and steps:
            kernel_ms, shell_ms = b.step_timing()
            return {"kernel_ms": kernel_ms, "shell_ms": None if shell_ms < 0 else shell_ms,
                "exchange_ms": sum(a.elapsed_time(c) for a, c in ev_comm) / len(ev_comm)}
        return None

    def fields(self):
        """(u, rho) of the local box incl. halos, host arrays in the reference layout"""
        return self.backend.download()

    def interior_to_global(self, local_arr, comps=1):
        """strip halos: returns the owned block and its global (x0,y0,z0)"""
        l = self.layout
        a = np.asarray(local_arr).reshape(comps, l.lN[2], l.lN[1], l.lN[0])
        s = [slice(h, n - h) for h, n in zip(l.H, l.lN)]
        return a[:, s[2], s[1], s[0]], tuple(c * (g // d) for c, g, d in zip(l.coord, l.gN, l.D))


class LocalGroup:
    """Several domains driven in lock-step inside ONE process (no torch.distributed): validates decomposition,
    shell/interior split and pack/unpack on a single GPU, or with CPU test doubles.  Halo buffers are copied directly
    between the members' tensors."""

    def __init__(self, global_N, D, nu, make_backend, overlap=True):
        n = D[0] * D[1] * D[2]
        self.sims = []
        for r in range(n):
            lay = DomainLayout(global_N, D, r)
            self.sims.append(DomainDecomposedLBM(global_N, D, nu, rank=r, backend=make_backend(lay), transport=self, overlap=overlap))

    def exchange(self, *a, **k):
        raise RuntimeError("LocalGroup exchanges in lock-step; use LocalGroup.run()")

    def _sync(self):
        for s in self.sims:
            s._join()

    def communicate_fi(self):
        lay0 = self.sims[0].layout
        for axis in lay0.split_axes():
            sent = [s.backend.extract(axis, getattr(s.backend, "comm", None)) for s in self.sims]
            self._sync()
            for s, (sp, sm) in zip(self.sims, sent):
                plus, minus = self.sims[s.layout.neighbor(axis, +1)], self.sims[s.layout.neighbor(axis, -1)]
                plus.backend.recv_buffers(axis)[1].copy_(sp)     # my + face -> its - halo
                minus.backend.recv_buffers(axis)[0].copy_(sm)    # my - face -> its + halo
            self._sync_all_devices()
            for s in self.sims:
                s.backend.insert(axis, getattr(s.backend, "comm", None))
            self._sync()
        if getattr(self.sims[0].backend, "thermal", False):
            for axis in lay0.split_axes():
                sent = [s.backend.extract_g(axis, getattr(s.backend, "comm", None)) for s in self.sims]
                self._sync()
                for s, (sp, sm) in zip(self.sims, sent):
                    plus, minus = self.sims[s.layout.neighbor(axis, +1)], self.sims[s.layout.neighbor(axis, -1)]
                    plus.backend.recv_buffers_g(axis)[1].copy_(sp)
                    minus.backend.recv_buffers_g(axis)[0].copy_(sm)
                self._sync_all_devices()
                for s in self.sims:
                    s.backend.insert_g(axis, getattr(s.backend, "comm", None))
                self._sync()

    def _sync_all_devices(self):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except ImportError:
            pass

    def initialize(self):
        for s in self.sims:
            s.backend.initialize(); s.backend.increment_time_step(1)
        self.communicate_fi()
        for s in self.sims:
            s.backend.reset_time_step(); s.initialized = True

    def run(self, steps):
        if not self.sims[0].initialized:
            self.initialize()
        for i in range(steps):
            wf = (i + 1 == steps)
            for s in self.sims:
                b, lay = s.backend, s.layout
                if s.overlap:
                    for box in lay.shell_boxes():
                        b.stream_collide(box, wf, b.comm)
                    b.stream_collide(lay.interior_box(), wf, b.compute)
                else:
                    b.stream_collide(lay.whole_box(), wf, getattr(b, "compute", None))
            self._sync()
            self.communicate_fi()
            for s in self.sims:
                s.backend.increment_time_step(1)

    def gather_u_rho(self):
        gN = self.sims[0].layout.gN
        u = np.zeros((3, gN[2], gN[1], gN[0]), np.float32); rho = np.zeros((1, gN[2], gN[1], gN[0]), np.float32)
        for s in self.sims:
            lu, lr = s.fields()
            ub, off = s.interior_to_global(lu, 3); rb, _ = s.interior_to_global(lr, 1)
            sl = (slice(None), slice(off[2], off[2] + ub.shape[1]), slice(off[1], off[1] + ub.shape[2]), slice(off[0], off[0] + ub.shape[3]))
            u[sl] = ub; rho[sl] = rb
        return u.ravel(), rho.ravel()
