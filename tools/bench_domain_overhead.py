#!/usr/bin/env python3
"""What one rank of an N-GPU run pays besides the interior kernel: boundary-shell launches, pack/unpack of the halo faces and
stream joins, measured on ONE GPU with a loopback transport (each face receives the rank's own opposite face = a periodic
single-domain problem, physically valid).  The xGMI transfer itself is the only thing missing.
usage: bench_domain_overhead.py [--D 4 2 1] [--size 512 512 512] [--steps 100] [--dtype f32|fp16c] [--no-overlap]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


class Loopback:
    def exchange(self, axis, send_p, send_m, recv_p, recv_m):
        recv_m.copy_(send_p, non_blocking=True); recv_p.copy_(send_m, non_blocking=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--D", type=int, nargs=3, default=[4, 2, 1]); ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--steps", type=int, default=100); ap.add_argument("--dtype", default="f32"); ap.add_argument("--no-overlap",
        action="store_true"); ap.add_argument("--phases", action="store_true"); ap.add_argument("--x-shell", type=int, default=0)
    a = ap.parse_args()
    import torch
    import latticeurbanwind_amd as luw
    from latticeurbanwind_amd.distributed import DomainDecomposedLBM
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import channel_state
    luw.load()
    if a.x_shell:
        from latticeurbanwind_amd.distributed import DomainLayout
        DomainLayout.X_SHELL = a.x_shell
    D = tuple(a.D); N = tuple(s * d for s, d in zip(a.size, D))
    sim = DomainDecomposedLBM(N, D, 1.48e-7, rank=0, transport=Loopback(), overlap=not a.no_overlap, fp16c=a.dtype == "fp16c", device=0)
    ox, oy, oz = sim.global_offset
    fl, u, rho = channel_state(sim.lNx, sim.lNy, sim.lNz, ox, oy, oz, *N)
    sim.set_fields(fl, u, rho)
    sim.initialize(); sim.run(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); k_ms = (sim.run(a.steps, timed=True) or {}).get("kernel_ms"); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    cells = a.size[0] * a.size[1] * a.size[2]
    print("D=%s local=%s overlap=%s: %.3f ms/step -> %.0f MLUPS per GPU (interior kernel %.3f ms)" % (D, (sim.lNx, sim.lNy, sim.lNz), sim.overlap,
        dt / a.steps * 1e3, cells * a.steps / dt / 1e6, k_ms or 0))
    if a.phases:   # serialised phases, one stream: what each piece costs on its own
        b, lay = sim.backend, sim.layout
        st = b.compute
        def timed(fn, reps=20):
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps): fn()
            e1.record(st); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
        print("  whole box        %.3f ms" % timed(lambda: b.stream_collide(lay.whole_box(), 0, st)))
        print("  interior box     %.3f ms  %s" % (timed(lambda: b.stream_collide(lay.interior_box(), 0, st)), lay.interior_box()))
        for bx in lay.shell_boxes():
            print("  shell %-28s %.3f ms" % (bx, timed(lambda: b.stream_collide(bx, 0, st))))
        for ax in lay.split_axes():
            print("  extract axis %d   %.3f ms" % (ax, timed(lambda: b.extract(ax, st))))
            print("  insert  axis %d   %.3f ms" % (ax, timed(lambda: b.insert(ax, st))))


if __name__ == "__main__":
    main()
