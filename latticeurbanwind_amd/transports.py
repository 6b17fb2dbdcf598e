"""How halo messages travel between the ranks of a decomposed run (latticeurbanwind_amd.distributed): RCCL point-to-point through torch.distributed, its
one-rank rehearsal (every neighbour is the rank itself), the peer-loopback twin without any copy, and the host-staged variant for gloo test worlds."""
from .layout import C19


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
