#!/usr/bin/env python3
"""GPU box: what a cell costs in boxes of different shapes of ONE haloed rank domain (the boxes a decomposed step launches, and alternatives), each box
alone on the device: time per launch (HIP events on the launch stream) -> ns per cell and the rate against the whole box.
usage: box_rate_probe.py <f32|fp16c> [Nx Ny Nz Dx Dy Dz]   (local owned block, default 512 512 512 of n_gpu = 4 2 1)
Under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum the per-dispatch counters of the same launches give the HBM traffic of each shape
(launch order = the order printed here, `reps` launches per box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latticeurbanwind_amd as luw
from latticeurbanwind_amd.distributed import DomainLayout
from bench import channel_state

fp16c = len(sys.argv) > 1 and sys.argv[1] == "fp16c"
v = [int(a) for a in sys.argv[2:8]] if len(sys.argv) >= 8 else [512, 512, 512, 4, 2, 1]
size, D = tuple(v[:3]), tuple(v[3:])
N = tuple(s * d for s, d in zip(size, D))
lay = DomainLayout(N, D, 0)
luw.load()
g = luw.LBM(*lay.lN, 1.48e-7, fp16c=fp16c, D=D, O=lay.O, device=0)
fl, u, rho = channel_state(*lay.lN, *lay.O, *N)
g.flags.data[:] = fl; g.u.data[:] = u; g.rho.data[:] = rho
g.run(0)
st = torch.cuda.Stream()
g.set_stream(st.cuda_stream)
w = lay.whole_box()
x0, x1, y0, y1, z0, z1 = w
boxes = [("whole", w), ("interior", lay.interior_box())] + [("shell%d" % k, b) for k, b in enumerate(lay.shell_boxes())]
q = (z1 - z0) // 4
boxes += [("z-quarter, whole rows", (x0, x1, y0, y1, z0, z0 + q)), ("x-half", (x0, x0 + (x1 - x0) // 2, y0, y1, z0, z1)),
          ("x 128..384 all y", (x0 + 128, x1 - 128, y0, y1, z0, z1)), ("x-slab 128 all y", (x0, x0 + 128, y0, y1, z0, z1)),
          ("x-slab 256 all y", (x0, x0 + 256, y0, y1, z0, z1)), ("y-half", (x0, x1, y0, y0 + (y1 - y0) // 2, z0, z1))]
reps = int(os.environ.get("PROBE_REPS", "10"))
# PROBE_XFACE=out: the step kernels also write the x faces of boxes that hold a border column (luw_set_x_face_buffers); =inout: and read the incoming ones
# from buffers instead of the lattice (luw_set_x_face_inputs before every launch): what the x-face instantiation costs a box
xmode = os.environ.get("PROBE_XFACE", "")
if xmode:
    A = g.area(0)
    xdt = torch.int16 if fp16c else torch.float32
    xbuf = [torch.zeros(5 * A, dtype=xdt, device="cuda") for _ in range(4)]
    g.set_x_face_buffers(xbuf[0].data_ptr(), xbuf[1].data_ptr())
    print("x-face mode:", xmode, flush=True)
def launch(b):
    if xmode == "inout": g.set_x_face_inputs(xbuf[2].data_ptr(), xbuf[3].data_ptr()); g.increment_time_step(1)
    g.enqueue_stream_collide(b, False)
    if xmode != "inout": g.increment_time_step(1)
base = None
for name, b in boxes:
    cells = (b[1] - b[0]) * (b[3] - b[2]) * (b[5] - b[4])
    if cells <= 0: continue
    for _ in range(2):
        launch(b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        launch(b)
    e1.record(st); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ns = ms * 1e6 / cells
    if base is None: base = ns
    print("%-26s box %-32s %11d cells  %.4f ms  %.5f ns/cell  %.3f of the whole-box rate" % (name, b, cells, ms, ns, base / ns), flush=True)
g.close()
