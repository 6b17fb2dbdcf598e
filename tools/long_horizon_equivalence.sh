#!/usr/bin/env bash
# GPU box, oracle/_ref/ travelling: at PRODUCTION horizons every realisation of the flow is equally far from every other.  The 512x512x128 deck of
# tools/e2e_wall.sh (one building, nudging + sponge, no VK inlet: deterministic), NSTEP steps of which the last PURGE are averaged, through the reference's
# FP32 and shipped builds and through this repo's driver (FP32; FP16C exact; FP16C native): RMSE of the TIME-AVERAGED velocity u_avg and of the final u between
# all pairs, lattice units.   usage: tools/long_horizon_equivalence.sh <out dir> [NSTEP] [PURGE]
set -u
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R"; O="$1"; NSTEP="${2:-3000}"; PURGE="${3:-1000}"; mkdir -p "$O"
W=$(mktemp -d)
python3 - "$W" "$NSTEP" "$PURGE" <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import make_refcases as mr
mr.write_case(sys.argv[1], "E", 1.0, ["enable_buffer_nudging = true", "enable_top_sponge = true", "sponge_thickness_m = 64"], dims=(1024, 1024, 192), building=True,
              nstep=int(sys.argv[2]), unsteady=0, purge=int(sys.argv[3]))
PY
for tag in ref_fp32 ref_shipped ours_fp32 ours_fp16c_exact ours_fp16c_native; do cp -r "$W/E" "$W/$tag"; done
( cd oracle/_ref && ./FluidX3D_fp32 "$W/ref_fp32/conf.luwpf" > "$W/ref_fp32/console.log" 2>&1 </dev/null ); echo "reference fp32 rc=$?"
( cd oracle/_ref && ./FluidX3D "$W/ref_shipped/conf.luwpf" > "$W/ref_shipped/console.log" 2>&1 </dev/null ); echo "reference shipped rc=$?"
latticeurbanwind_amd/host/luw_driver "$W/ours_fp32/conf.luwpf" --ddf fp32 > "$W/ours_fp32/console.log" 2>&1; echo "driver fp32 rc=$?"
latticeurbanwind_amd/host/luw_driver "$W/ours_fp16c_exact/conf.luwpf" --ddf fp16c --arith exact > "$W/ours_fp16c_exact/console.log" 2>&1; echo "driver fp16c exact rc=$?"
latticeurbanwind_amd/host/luw_driver "$W/ours_fp16c_native/conf.luwpf" --ddf fp16c --arith native > "$W/ours_fp16c_native/console.log" 2>&1; echo "driver fp16c native rc=$?"
python3 - "$W" "$NSTEP" "$PURGE" <<'PY' | tee "$O/long_horizon.txt"
import sys, os, glob, itertools
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
from vtkio import read_vtk
W, K, P = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
fac = np.float32(5.0) / np.float32(0.1)
tags = ("ref_fp32", "ref_shipped", "ours_fp32", "ours_fp16c_exact", "ours_fp16c_native")
F = {}
for t in tags:
    h, a = read_vtk(glob.glob(os.path.join(W, t, "RESULTS", "vtk", "*_avg-%09d.vtk" % K))[0])
    F[t] = {"u_avg": a["u_avg"] / fac, "fluid": a["fluid"][..., 0] != 0, "u": read_vtk(glob.glob(os.path.join(W, t, "RESULTS", "vtk", "*_raw_u-%09d.vtk" % K))[0])[1]["data"] / fac}
fl = F["ref_fp32"]["fluid"]
print("512x512x128 deck (one building, nudging + sponge), %d steps, the last %d averaged; %d non-solid cells; lattice units (u_lbm = 0.1 at the profile's maximum)" % (K, P, int(fl.sum())))
print("mean |u_avg| per run: " + "  ".join("%s %.5f" % (t, float(np.sqrt((F[t]["u_avg"][fl].astype(np.float64) ** 2).sum(-1)).mean())) for t in tags))
print("%-40s %12s %12s" % ("pair", "u_avg RMSE", "final u RMSE"))
for a, b in itertools.combinations(tags, 2):
    r = lambda k: float(np.sqrt((((F[a][k] - F[b][k])[fl].astype(np.float64)) ** 2).sum(-1).mean()))
    print("%-40s %12.3e %12.3e" % (a + " vs " + b, r("u_avg"), r("u")))
PY
rm -rf "$W"
