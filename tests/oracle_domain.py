"""CPU test double for latticeurbanwind_amd.distributed: one domain backed by the oracle, halo buffers as torch CPU
tensors (so the real DomainDecomposedLBM / TorchDistTransport code runs under gloo).  tests/ only."""
import numpy as np
import torch

from oracle import oracle
from latticeurbanwind_amd.distributed import C19


class OracleDomain:
    def __init__(self, layout, nu, fp16c=False, alpha=None):
        self.layout = layout
        self.thermal = alpha is not None
        self.o = oracle.OracleLBM(*layout.lN, nu, fp16c=fp16c, D=layout.D, O=layout.O, alpha=alpha)
        self.np_dtype = np.uint16 if fp16c else np.float32
        self.t_dtype = torch.float16 if fp16c else torch.float32   # FP16C codes ride as float16 bit patterns (NCCL has no int16)
        self.buf = {a: [torch.zeros(5 * self.o.area(a), dtype=self.t_dtype) for _ in range(4)] for a in layout.split_axes()}
        self.gbuf = {a: [torch.zeros(self.o.area(a), dtype=self.t_dtype) for _ in range(4)] for a in layout.split_axes()} if self.thermal else {}

    def set_fields(self, flags, u, rho, T=None):
        self.o.flags[:] = flags; self.o.u[:] = u; self.o.rho[:] = rho
        if T is not None: self.o.T[:] = T

    def initialize(self): self.o.initialize()
    def get_t(self): return self.o.t
    def increment_time_step(self, n=1): self.o.t += n
    def reset_time_step(self): self.o.t = 0

    def stream_collide(self, box, write_fields, stream):
        assert tuple(box) == tuple(self.layout.whole_box())      # the test double always does the whole domain at once
        if self.thermal: self.o.stream_collide_thermal()
        else: self.o.stream_collide()

    def extract(self, axis, stream):
        bp, bm = self.o.extract_fi(axis)
        self.buf[axis][0].copy_(torch.from_numpy(bp.view(np.float16) if self.np_dtype == np.uint16 else bp))
        self.buf[axis][1].copy_(torch.from_numpy(bm.view(np.float16) if self.np_dtype == np.uint16 else bm))
        return self.buf[axis][0], self.buf[axis][1]

    def recv_buffers(self, axis): return self.buf[axis][2], self.buf[axis][3]

    def insert(self, axis, stream):
        rp = self.buf[axis][2].numpy().view(self.np_dtype); rm = self.buf[axis][3].numpy().view(self.np_dtype)
        self.o.insert_fi(axis, np.ascontiguousarray(rp), np.ascontiguousarray(rm))

    def _view(self, a): return a.view(np.float16) if self.np_dtype == np.uint16 else a

    def extract_g(self, axis, stream):
        bp, bm = self.o.extract_gi(axis)
        self.gbuf[axis][0].copy_(torch.from_numpy(self._view(bp))); self.gbuf[axis][1].copy_(torch.from_numpy(self._view(bm)))
        return self.gbuf[axis][0], self.gbuf[axis][1]

    def recv_buffers_g(self, axis): return self.gbuf[axis][2], self.gbuf[axis][3]

    def insert_g(self, axis, stream):
        rp = self.gbuf[axis][2].numpy().view(self.np_dtype); rm = self.gbuf[axis][3].numpy().view(self.np_dtype)
        self.o.insert_gi(axis, np.ascontiguousarray(rp), np.ascontiguousarray(rm))

    # ---- edge messages of the one-phase exchange, restated from the slot algebra of the reference's transfer kernels (FX/kernel.cpp:2241-2270: an odd
    # population i is read at the +c_i neighbour of the face cell in plane (t odd ? i+1 : i), an even one at the cell itself in plane (t odd ? i-1 : i);
    # inserts mirror that) for the ONE population that crosses both cuts of an edge, on the line of cells where the two faces meet
    def _edge_line(self, e, sender):
        i = 7 + e
        c, N, odd = C19[i], self.layout.lN, bool(i & 1)
        t_odd = self.o.t & 1
        io = i if odd else i - 1
        plane = (io + 1 if t_odd else io) if odd else (io if t_odd else io + 1)
        idx = []
        for a in range(3):
            if c[a] == 0: idx.append(np.arange(N[a]))
            elif odd: idx.append((N[a] - 1 if c[a] > 0 else 0) if sender else (1 if c[a] > 0 else N[a] - 2))
            else: idx.append((N[a] - 2 if c[a] > 0 else 1) if sender else (0 if c[a] > 0 else N[a] - 1))
        n = idx[0] + (idx[1] + idx[2] * N[1]) * N[0]
        return plane * self.o.N + n

    def extract_edges(self, stream):
        if not hasattr(self, "ebuf"):
            self.ebuf = {e: [torch.zeros(self.layout.edge_length(e), dtype=self.t_dtype) for _ in range(2)] for e in self.layout.edges()}
        for e, b in self.ebuf.items():
            b[0].copy_(torch.from_numpy(self._view(self.o.fi[self._edge_line(e, True)].copy())))
        return [(e, b[0], b[1]) for e, b in sorted(self.ebuf.items())]

    def insert_edges(self, stream):
        import os
        if os.environ.get("LUW_TEST_DROP_EDGES") == "1": return            # tests/test_distributed_gloo.py: the negative control
        for e, b in getattr(self, "ebuf", {}).items():
            self.o.fi[self._edge_line(e, False)] = b[1].numpy().view(self.np_dtype)

    def download_T(self): return self.o.T

    def download(self): return self.o.u, self.o.rho
