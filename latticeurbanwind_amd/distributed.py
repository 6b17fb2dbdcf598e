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

# D3Q19 directions c_i (FX/kernel.cpp:890-893); edge message e = 0..11 carries population 7 + e to the domain in direction c_(7+e)
C19 = ((0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 0), (-1, -1, 0), (1, 0, 1), (-1, 0, -1), (0, 1, 1), (0, -1, -1),
       (1, -1, 0), (-1, 1, 0), (1, 0, -1), (-1, 0, 1), (0, 1, -1), (0, -1, 1))


def choose_decomposition(world, split_x=False):
    """n_gpu and per-GPU lattice shape factors for a weak-scaled tile of 512^3 cells per GPU.

    split_x=False (default): the memory-fastest axis is kept whole -- rows stay complete memory lines, the y/z boundary
    shells are whole rows and the halo traffic hides behind the interior: 8 GPUs cover the 2048x1024x512 tile of
    BASELINE configs[3] as n_gpu=[1,4,2] (local 2048x256x256: the least halo area among the x-whole grids, and the fastest
    rank step measured, 3.42 ms vs 3.55 ms for [1,2,4]; tools/bench_layouts.py).  split_x=True reproduces the deck's literal
    n_gpu=[4,2,1] (local 512^3); with x split the step runs the whole box first and exchanges afterwards (measured on
    MI355X, one rank with loopback halos: 3.77 ms sequential vs 4.2-5.2 ms with an x shell, vs 3.37 ms undivided).
    Returns (D, global_lattice)."""
    if split_x:
        table = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (4, 2, 1), 16: (4, 4, 1)}
    else:
        table = {1: (1, 1, 1), 2: (1, 2, 1), 4: (1, 2, 2), 8: (1, 4, 2), 16: (1, 4, 4)}
    if world in table:
        return table[world]
    d = [1, 1, 1]
    n, ax = world, 0
    for p in (2, 3, 5, 7):
        while n % p == 0:
            d[(ax % 2) + 1 if not split_x else ax % 3] *= p; ax += 1; n //= p
    if n != 1:
        d[1 if not split_x else 0] *= n
    return tuple(d)


def tile_lattice(world):
    """global lattice of the weak-scaled benchmark tile: 512^3 cells per GPU, growing x, then y, then x again
    (1: 512^3 = BASELINE configs[1]; 2: 1024x512x512; 4: 1024x1024x512; 8: 2048x1024x512 = configs[3])"""
    g = [512, 512, 512]
    n, ax = world, 0
    while n > 1 and n % 2 == 0:
        g[(0, 1)[ax % 2]] *= 2; ax += 1; n //= 2
    g[2] *= n
    return tuple(g)


class DomainLayout:
    """Pure host logic: where a rank sits, what it owns, whom it talks to (FX/lbm.cpp:1066-1073,1912-1931)."""

    X_SHELL = 128  # thickness of the x boundary slabs (see shell_boxes; csrc/luw_group.hpp group_x_shell)

    def __init__(self, global_N, D, rank, x_shell=None):
        if x_shell: self.X_SHELL = int(x_shell)
        self.gN = tuple(int(v) for v in global_N)
        self.D = tuple(int(v) for v in D)
        Dx, Dy, Dz = self.D
        if any(g % d for g, d in zip(self.gN, self.D)):
            raise ValueError("LBM grid %s is not equally divisible in domains %s" % (self.gN, self.D))  # FX/lbm.cpp:1058-1059 shrinks; we refuse
        if not 0 <= rank < Dx * Dy * Dz:
            raise ValueError("rank outside the domain grid")
        self.rank = rank
        self.coord = ((rank % (Dx * Dy)) % Dx, (rank % (Dx * Dy)) // Dx, rank // (Dx * Dy))
        self.H = tuple(int(d > 1) for d in self.D)                         # halo offsets
        self.lN = tuple(g // d + 2 * h for g, d, h in zip(self.gN, self.D, self.H))
        self.O = tuple(c * (g // d) - h for c, g, d, h in zip(self.coord, self.gN, self.D, self.H))

    def rank_of(self, coord):
        x, y, z = coord
        return x + (y + z * self.D[1]) * self.D[0]

    def neighbor(self, axis, sign):
        c = list(self.coord)
        c[axis] = (c[axis] + sign) % self.D[axis]
        return self.rank_of(c)

    def neighbor_dir(self, c):
        """rank of the domain in direction c = (cx, cy, cz), periodic"""
        return self.rank_of(tuple((k + d) % n for k, d, n in zip(self.coord, c, self.D)))

    def split_axes(self):
        return [a for a in range(3) if self.D[a] > 1]

    def edges(self):
        """edge messages this domain takes part in: population 7 + e crosses two cuts, both of them split"""
        return [e for e in range(12) if all(self.D[a] > 1 for a in range(3) if C19[7 + e][a])]

    def edge_length(self, e):
        """cells of edge e's line: the local extent of the axis its population does not move along"""
        return self.lN[[a for a in range(3) if C19[7 + e][a] == 0][0]]

    # ---- boxes (x0,x1,y0,y1,z0,z1) in local coordinates: ONE implementation, the library's (luw_step_boxes, csrc/luw_group.hpp: pure host arithmetic, also
    # what the one-process host luw_group_* cuts its domains with).  whole = the non-halo cells; interior + the disjoint shell slabs cover it exactly once;
    # y, z slabs are the one cell layer next to a halo (whole rows), x slabs whole blocks of X_SHELL cells from the first owned cell on.
    def _boxes(self):
        key = (self.lN, self.H, self.X_SHELL)
        if getattr(self, "_box_key", None) != key:
            import ctypes as C
            from . import capi
            u3 = C.c_uint32 * 3
            whole, inner, shell, n, ok = (C.c_uint32 * 6)(), (C.c_uint32 * 6)(), (C.c_uint32 * 36)(), C.c_uint32(0), C.c_int(0)
            capi.check(capi.load().luw_step_boxes(u3(*self.lN), u3(*self.H), int(self.X_SHELL), whole, inner, shell, C.byref(n), C.byref(ok)))
            self._box_key = key
            self._box_val = (tuple(whole), tuple(inner), [tuple(shell[6 * k:6 * k + 6]) for k in range(n.value)], bool(ok.value))
        return self._box_val

    def whole_box(self): return self._boxes()[0]
    def interior_box(self): return self._boxes()[1]
    def shell_boxes(self): return list(self._boxes()[2])
    def can_overlap(self): return self._boxes()[3]       # every split axis has at least four owned layers

    def local_slices(self):
        """slices of the GLOBAL (z,y,x) array that fill the local box incl. halos (periodic wrap), as index arrays"""
        idx = []
        for a in range(3):
            idx.append((np.arange(self.lN[a]) + self.O[a]) % self.gN[a])
        return idx  # x, y, z index arrays


class TorchDistTransport:
    """halo swap over torch.distributed point-to-point ops (RCCL on GPUs)"""

    def __init__(self, layout, group=None):
        import torch.distributed as dist
        self.dist, self.layout, self.group = dist, layout, group

    def exchange(self, axis, send_p, send_m, recv_p, recv_m):
        """send_p -> +neighbour (arrives as its recv_m); send_m -> -neighbour (its recv_p)."""
        dist = self.dist
        plus, minus = self.layout.neighbor(axis, +1), self.layout.neighbor(axis, -1)
        # fixed issue order on every rank keeps the pairing unambiguous when plus == minus (D = 2)
        ops = [dist.P2POp(dist.isend, send_p, plus, self.group), dist.P2POp(dist.isend, send_m, minus, self.group),
               dist.P2POp(dist.irecv, recv_m, minus, self.group), dist.P2POp(dist.irecv, recv_p, plus, self.group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def exchange_all(self, messages):
        """ONE batch for everything a step moves.  messages: (send, recv, c) in a fixed order of message types -- `send` leaves for the domain in direction c,
        `recv` takes the same type of message from the domain in direction -c.  Every rank lists the types in the same order, so the k-th send of A to B is
        the k-th receive of B from A also where several directions lead to the same rank (two domains along an axis; a rank that is its own neighbour)."""
        dist, lay = self.dist, self.layout
        if not messages:         # a domain without a cut: nothing to move
            return
        ops = [dist.P2POp(dist.isend, s, lay.neighbor_dir(c), self.group) for s, _, c in messages]
        ops += [dist.P2POp(dist.irecv, r, lay.neighbor_dir(tuple(-v for v in c)), self.group) for _, r, c in messages]
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def warm_up(self, device, dtype=None, measure=0):
        """One full-size exchange per split axis on scratch buffers, BEFORE the lattice is allocated: RCCL builds its
        point-to-point connections (channels, staging buffers) at the first send/recv to a peer.  Measured on MI355X
        (tools/check_nccl_self.py, LUW_SELF_EARLY): when that set-up happens in a process that has already allocated and freed
        lattice-sized arrays, the interior kernel that follows runs 24 % slower for the life of the solver (2048x258x258 FP32:
        4.44 instead of 3.55 ms); with the connections built first it does not.  Full-size messages, so that every channel the
        real faces will use is connected now; the scratch buffers go back to torch's caching allocator, from which the domain's
        halo buffers of the same sizes are then served."""
        import torch
        lN = self.layout.lN
        wire = {}
        for a in self.layout.split_axes():
            A = lN[(a + 1) % 3] * lN[(a + 2) % 3]
            bufs = [torch.zeros(5 * A, dtype=dtype or torch.float32, device=device) for _ in range(4)]
            self.exchange(a, *bufs)
            if measure:
                # the wire alone: `measure` more exchanges of the same faces, HIP events on the stream the transport enqueues on
                torch.cuda.synchronize(device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(measure):
                    self.exchange(a, *bufs)
                e1.record(); e1.synchronize()
                ms = e0.elapsed_time(e1) / measure
                out_bytes = 2 * bufs[0].numel() * bufs[0].element_size()       # the + face and the - face leave, as many bytes arrive
                wire["xyz"[a]] = {"bytes_out": out_bytes, "ms": round(ms, 4), "GBps_out": round(out_bytes / (ms * 1e-3) / 1e9, 2) if ms > 0 else None,
                                  "to_ranks": [self.layout.neighbor(a, +1), self.layout.neighbor(a, -1)]}
        # the one-phase exchange also talks to the diagonal neighbours (edge messages): one batch of everything, so that those connections exist now as well
        edges = self.layout.edges() if hasattr(self.layout, "edges") else []
        if edges:
            unit = lambda a, sgn: tuple(sgn if k == a else 0 for k in range(3))
            msgs = []
            for a in self.layout.split_axes():
                A = lN[(a + 1) % 3] * lN[(a + 2) % 3]
                msgs += [(torch.zeros(5 * A, dtype=dtype or torch.float32, device=device), torch.zeros(5 * A, dtype=dtype or torch.float32, device=device),
                          unit(a, sgn)) for sgn in (+1, -1)]
            msgs += [(torch.zeros(self.layout.edge_length(e), dtype=dtype or torch.float32, device=device),
                      torch.zeros(self.layout.edge_length(e), dtype=dtype or torch.float32, device=device), C19[7 + e]) for e in edges]
            self.exchange_all(msgs)
        torch.cuda.synchronize(device)
        return wire


class SelfExchangeTransport(TorchDistTransport):
    """every neighbour is THIS rank: each face leaves and comes back through the real transport's self send / receive (RCCL on a GPU box).
    Physically the rank's block made periodic.  What one rank of an N-GPU run does per step -- boundary shell, pack, exchange, unpack, interior,
    pipelining -- on one GPU, without the wire to another device (bench.py's rank-shape blocks, tests/rank_shape_worker.py)."""

    def __init__(self, layout, group=None):
        super().__init__(layout, group)
        import torch.distributed as dist
        me = dist.get_rank() if dist.is_initialized() else 0
        self.layout = type("SelfNeighbours", (), {"neighbor": staticmethod(lambda axis, sign: me), "neighbor_dir": staticmethod(lambda c: me),
            "lN": layout.lN, "split_axes": layout.split_axes, "edges": layout.edges, "edge_length": layout.edge_length})()


class PeerLoopbackTransport:
    """every neighbour is THIS rank and no transport at all: the face buffers a step fills ARE the buffers its unpack reads (HipDomain(alias_faces=True):
    the + face is written where "what came from the - side" is read, and vice versa) -- what the one-process host's peer stores do between two domains
    (csrc/luw_group.hpp: the pack kernel, or the step kernels themselves for the x faces, write straight into the neighbour's receive buffer), with the
    rank as its own neighbour.  Physically the rank's block made periodic, like SelfExchangeTransport, minus RCCL's copy kernels: bench.py's rank-shape
    blocks carry both, so that what a rank pays for the transport is on the line."""
    alias_faces = True

    def __init__(self, layout):
        self.layout = layout

    def exchange(self, axis, send_p, send_m, recv_p, recv_m):
        assert send_p.data_ptr() == recv_m.data_ptr() and send_m.data_ptr() == recv_p.data_ptr()

    def exchange_all(self, messages):
        assert all(s.data_ptr() == r.data_ptr() for s, r, _ in messages)

    def warm_up(self, device, dtype=None, measure=0):
        return {}


def init_rccl_process_group(local_rank, timeout=None):
    """`torch.distributed` over RCCL for one process per GPU.  RCCL's point-to-point kernels are launched while the interior
    collide-stream kernel fills every CU, so the process group's internal stream is asked to be a high-priority one (like the
    communication stream of `HipDomain`): the halo copies are dispatched ahead of the interior's remaining workgroups instead
    of behind them (LUW_COMM_PRIORITY=0 turns both off for A/B runs, LUW_NCCL_PRIORITY=0 this one alone)."""
    import os
    import torch
    import torch.distributed as dist
    kw = {}
    if os.environ.get("LUW_NCCL_PRIORITY", os.environ.get("LUW_COMM_PRIORITY", "1")) != "0":
        try:
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
            kw["pg_options"] = opts
        except Exception:           # a torch build without the option: default stream priority
            pass
    if timeout is not None:
        kw["timeout"] = timeout
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)


class HostStagedTransport(TorchDistTransport):
    """the same swap staged through host memory, for process groups that cannot move device memory (gloo): lets several
    ranks share ONE GPU in tests; never the production path"""

    def exchange(self, axis, send_p, send_m, recv_p, recv_m):
        import torch
        sp, sm = send_p.cpu(), send_m.cpu()                     # ordered after the pack kernel on the current stream
        rp, rm = torch.empty_like(sp), torch.empty_like(sm)
        super().exchange(axis, sp, sm, rp, rm)
        recv_p.copy_(rp); recv_m.copy_(rm)

    def exchange_all(self, messages):
        import torch
        staged = [(s.cpu(), torch.empty(r.shape, dtype=r.dtype), c) for s, r, c in messages]
        super().exchange_all(staged)
        for (_, r, _), (_, h, _) in zip(messages, staged):
            r.copy_(h)


class HipDomain:
    """One LBM domain on one GPU through the C-ABI; buffers are torch CUDA tensors, work is enqueued on torch streams."""

    def __init__(self, layout, nu, fp16c=False, kernel=0, device=0, alias_faces=False, **kw):
        import torch
        from .lbm import LBM
        self.torch = torch
        self.layout = layout
        self.device = torch.device("cuda", device)
        self.lbm = LBM(*layout.lN, nu, fp16c=fp16c, D=layout.D, O=layout.O, device=device, kernel=kernel, **kw)
        # halo buffers travel as raw values: FP32 as float32, FP16C codes as float16 bit patterns (RCCL/NCCL has no int16 type;
        # point-to-point ops copy bytes, nothing interprets the halves)
        self.dtype = torch.float16 if fp16c else torch.float32
        self.compute = torch.cuda.Stream(device=self.device)
        # the boundary shell, the halo pack/unpack kernels and the exchange run on a high-priority queue so that they are not
        # stuck behind the interior kernel's workgroups (LUW_COMM_PRIORITY=0 turns that off for A/B runs)
        import os
        self.comm = torch.cuda.Stream(device=self.device, priority=(-1 if os.environ.get("LUW_COMM_PRIORITY", "1") != "0" else 0))
        # the schedule of a step -- which box on which stream, behind which event -- is the library's (luw_domain_step_*, shared with luw_group_*)
        self.step = self.lbm.domain_step_create(self.compute.cuda_stream, self.comm.cuda_stream, layout.X_SHELL)
        self.thermal = kw.get("alpha") is not None       # thermal D3Q7 lattice: one more population per face cell travels
        self.buf, self.gbuf = {}, {}
        self.x_insert_fused, self.x_pairs = False, None
        for a in layout.split_axes():
            A = self.lbm.area(a)
            self.buf[a] = [torch.zeros(5 * A, dtype=self.dtype, device=self.device) for _ in range(4)]  # send_p, send_m, recv_p, recv_m
            if alias_faces:      # PeerLoopbackTransport: the rank is its own neighbour and the faces are written where they are read
                self.buf[a][0], self.buf[a][1] = self.buf[a][3], self.buf[a][2]
            if a == 0 and os.environ.get("LUW_X_FACE_FUSED", "1") != "0":
                # the step kernels that hold the first / last owned x column write the x faces into the send buffers themselves; extract(0) then has
                # nothing to launch (LUW_X_FACE_FUSED=0: the pack kernel as before, A/B switch) ...
                self.lbm.set_x_face_buffers(self.buf[0][0].data_ptr(), self.buf[0][1].data_ptr())
                # ... and read the x faces they receive from the receive buffers (insert_deferred; LUW_X_INSERT_FUSED=0: the unpack kernel, A/B switch)
                self.x_insert_fused = os.environ.get("LUW_X_INSERT_FUSED", "1") != "0"
                if alias_faces and self.x_insert_fused:
                    # written where they are read: the step that reads one pair of buffers writes the other (a neighbour's stores would race with the loads)
                    other = [torch.zeros(5 * A, dtype=self.dtype, device=self.device) for _ in range(2)]
                    self.x_pairs = [self.buf[0], [other[1], other[0], other[0], other[1]]]        # send_p is recv_m, send_m is recv_p
            if self.thermal:
                self.gbuf[a] = [torch.zeros(A, dtype=self.dtype, device=self.device) for _ in range(4)]
                if alias_faces:
                    self.gbuf[a][0], self.gbuf[a][1] = self.gbuf[a][3], self.gbuf[a][2]
        # edge messages of the one-phase exchange: [send, receive] per edge, one element per cell of the third axis
        self.ebuf = {e: [torch.zeros(self.lbm.edge_length(e), dtype=self.dtype, device=self.device) for _ in range(2)] for e in layout.edges()}
        if alias_faces:
            for e in self.ebuf: self.ebuf[e][0] = self.ebuf[e][1]

    # host fields (reference layout, local box incl. halos)
    def set_fields(self, flags, u, rho, T=None):
        self.lbm.flags.data[:] = flags; self.lbm.u.data[:] = u; self.lbm.rho.data[:] = rho
        if T is not None:
            self.lbm.T.data[:] = T

    def initialize(self):
        self.lbm.run(0)

    def get_t(self): return self.lbm.get_t()
    def increment_time_step(self, n=1): self.lbm.increment_time_step(n)
    def reset_time_step(self): self.lbm.reset_time_step()

    def stream_collide(self, box, write_fields, stream, sample=False):
        self.lbm.set_stream(stream.cuda_stream)
        self.lbm.enqueue_stream_collide(box, write_fields, sample)

    def stats_begin_sample(self): return self.lbm.stats_begin_sample()
    def configure_step(self, overlap):
        if self.lbm.domain_step_overlaps(self.step) != bool(overlap):
            self.lbm.domain_step_destroy(self.step)
            self.step = self.lbm.domain_step_create(self.compute.cuda_stream, self.comm.cuda_stream, self.layout.X_SHELL, overlap=overlap)
    def step_overlaps(self): return self.lbm.domain_step_overlaps(self.step)
    def step_launch(self, write_fields, timed=False): self.lbm.domain_step_launch(self.step, write_fields, timed)
    def step_separate_stats(self): self.lbm.domain_step_separate_stats(self.step)
    def step_timing(self): return self.lbm.domain_step_timing(self.step)

    def extract(self, axis, stream):
        self.lbm.set_stream(stream.cuda_stream)
        b = self.buf[axis]
        self.lbm.enqueue_extract_fi(axis, b[0].data_ptr(), b[1].data_ptr())
        return b[0], b[1]

    def recv_buffers(self, axis):
        return self.buf[axis][2], self.buf[axis][3]

    def insert(self, axis, stream):
        self.lbm.set_stream(stream.cuda_stream)
        b = self.buf[axis]
        self.lbm.enqueue_insert_fi(axis, b[2].data_ptr(), b[3].data_ptr())

    def insert_deferred(self, axis, stream):
        """insert(0) without its kernel where the library can do that (one-phase exchange: nothing packs from the lattice before the next step): the next
        step's kernels read the x faces from the receive buffers.  Other axes: the unpack kernel."""
        if axis != 0 or not self.x_insert_fused:
            return self.insert(axis, stream)
        self.lbm.set_stream(stream.cuda_stream)
        b = self.buf[0]
        self.lbm.set_x_face_inputs(b[2].data_ptr(), b[3].data_ptr())
        self.next_x_buffers()

    def next_x_buffers(self):
        """the x buffers of the next step: the other set (the set just filled is read while that step runs)"""
        if self.x_pairs:
            self.buf[0] = self.x_pairs[1] if self.buf[0] is self.x_pairs[0] else self.x_pairs[0]
            self.lbm.set_x_face_buffers(self.buf[0][0].data_ptr(), self.buf[0][1].data_ptr())

    def extract_edges(self, stream, only=None):
        """packs every edge (only: those edges); returns [(e, send, receive)]"""
        todo = sorted(e for e in self.ebuf if only is None or e in only)
        if todo:
            self.lbm.set_stream(stream.cuda_stream)
            self.lbm.enqueue_edges([self.ebuf[e][0].data_ptr() if e in todo else 0 for e in range(12)], insert=False)
        return [(e, self.ebuf[e][0], self.ebuf[e][1]) for e in todo]

    def insert_edges(self, stream, only=None):
        todo = [e for e in self.ebuf if only is None or e in only]
        if todo:
            self.lbm.set_stream(stream.cuda_stream)
            self.lbm.enqueue_edges([self.ebuf[e][1].data_ptr() if e in todo else 0 for e in range(12)], insert=True)

    def extract_g(self, axis, stream):
        self.lbm.set_stream(stream.cuda_stream)
        b = self.gbuf[axis]
        self.lbm.enqueue_extract_gi(axis, b[0].data_ptr(), b[1].data_ptr())
        return b[0], b[1]

    def recv_buffers_g(self, axis):
        return self.gbuf[axis][2], self.gbuf[axis][3]

    def insert_g(self, axis, stream):
        self.lbm.set_stream(stream.cuda_stream)
        b = self.gbuf[axis]
        self.lbm.enqueue_insert_gi(axis, b[2].data_ptr(), b[3].data_ptr())

    def download_T(self):
        self.torch.cuda.synchronize(self.device)
        self.lbm.T.read_from_device()
        return self.lbm.T.data

    def download(self):
        self.torch.cuda.synchronize(self.device)
        self.lbm.u.read_from_device(); self.lbm.rho.read_from_device()
        return self.lbm.u.data, self.lbm.rho.data

    # ---- what a deck run loop needs besides the step (inlet, probes, statistics)
    def set_coriolis(self, ox, oy, oz): self.lbm.set_coriolis(ox, oy, oz)

    def vk_attach(self, cell, face, point_data, mode_data, mode_count, stride, interp):
        self.lbm.vk_inlet_attach(cell, face, point_data, mode_data, mode_count, stride, interp)

    def vk_apply(self, stream):
        self.lbm.set_stream(stream.cuda_stream)
        self.lbm.vk_inlet_apply()

    def gather_attach(self, cells): self.lbm.gather_attach(cells)

    def gather_u(self):
        self.lbm.set_stream(self.compute.cuda_stream)
        return self.lbm.gather_u()

    def stats_reset(self): self.lbm.stats_reset()

    def stats_accumulate(self):
        self.lbm.set_stream(self.compute.cuda_stream)
        self.lbm.stats_accumulate()
        self.compute.synchronize()

    def stats_enqueue(self, stream):
        """the Welford update on `stream`, no host synchronisation (sampled steps inside DomainDecomposedLBM.run)"""
        self.lbm.set_stream(stream.cuda_stream)
        self.lbm.stats_accumulate()

    def stats_download(self):
        self.lbm.set_stream(self.compute.cuda_stream)
        return self.lbm.stats_download()

    def stats_download_T(self):
        self.lbm.set_stream(self.compute.cuda_stream)
        return self.lbm.stats_download_T()

    def close(self):
        self.torch.cuda.synchronize(self.device)
        if getattr(self, "step", None):
            self.lbm.domain_step_destroy(self.step); self.step = None
        self.lbm.close()


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
