#!/usr/bin/env bash
# u-field RMSE against the REAL reference at a bench-class size (GPU box, via gpurun): the same 512x512x128 profile deck (one
# building, nudging + sponge on) runs K steps on the reference's FP32 build and on this repo's driver (--ddf fp32), and on the
# shipped FP16C build vs --ddf fp16c; prints sqrt(mean |u_ours - u_ref|^2) over the non-solid cells in lattice units.
# usage: tools/e2e_rmse.sh [K]      (LUW_E2E_ARITH=exact|native: the FP16C run's arithmetic, default native like the driver's)
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"; cd "$R"
K="${1:-100}"
W=$(mktemp -d)
python3 - "$W" "$K" <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import make_refcases as mr
mr.write_case(sys.argv[1], "R", 1.0, ["enable_buffer_nudging = true", "enable_top_sponge = true", "sponge_thickness_m = 64"], dims=(1024, 1024, 192), building=True,
              nstep=int(sys.argv[2]), unsteady=0, purge=0)
PY
for tag in ref_fp32 ref_shipped ours_fp32 ours_fp16c; do cp -r "$W/R" "$W/$tag"; done
( cd oracle/_ref && ./FluidX3D_fp32 "$W/ref_fp32/conf.luwpf" > "$W/ref_fp32/console.log" 2>&1 </dev/null )
( cd oracle/_ref && ./FluidX3D "$W/ref_shipped/conf.luwpf" > "$W/ref_shipped/console.log" 2>&1 </dev/null )
latticeurbanwind_amd/host/luw_driver "$W/ours_fp32/conf.luwpf" --ddf fp32 > "$W/ours_fp32/console.log" 2>&1
latticeurbanwind_amd/host/luw_driver "$W/ours_fp16c/conf.luwpf" --ddf fp16c --arith "${LUW_E2E_ARITH:-native}" > "$W/ours_fp16c/console.log" 2>&1
python3 - "$W" "$K" <<'PY'
import sys, os, glob
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
from vtkio import read_vtk
W, K = sys.argv[1], int(sys.argv[2])
def load(tag):
    f = glob.glob(os.path.join(W, tag, "RESULTS", "vtk", "*_raw_u-%09d.vtk" % K))[0]
    h, d = read_vtk(f); return d["data"]
fac = 5.0 / 0.1     # si_ref_u = max profile U = 5 m/s at u_lbm = 0.1 -> SI per lattice unit
for a, b, label in (("ours_fp32", "ref_fp32", "FP32 DDFs"), ("ours_fp16c", "ref_shipped", "FP16C DDFs (shipped reference build)")):
    ua, ub = load(a), load(b)
    solid = ~np.isfinite(ub).all(-1) | ((ub == 0).all(-1) & (ua == 0).all(-1))
    d = ((ua - ub) / fac)[~solid].astype(np.float64)
    print("%-40s K=%d  cells %d  u RMSE %.3e  max |du| %.3e  (lattice units)" % (label, K, d.shape[0], np.sqrt((d ** 2).sum(-1).mean()), np.abs(d).max()))
PY
rm -rf "$W"
