"""One LBM domain on one GPU behind the interface latticeurbanwind_amd.distributed.DomainDecomposedLBM steps: the C-ABI solver, its two streams, its halo
buffers (torch CUDA tensors), and the calls that pack, hand over and unpack them."""


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
