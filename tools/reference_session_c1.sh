#!/usr/bin/env bash
# Fixture-regeneration session on the GPU box (needs oracle/_ref/ to travel: its .gpurunignore line off for this call, tests/golden/README.md):
# BASELINE configs[0] as a deck (tests/golden/refcases/CaseC1: 128^3, K = 100) through BOTH builds of the real reference and through this repo's driver
# (FP32; FP16C exact and native); writes the three mid-planes of u of each reference build at K = 100 (+ of u at K = 50 and u_avg) with global statistics to
# <out>/ref_{fp32,shipped}_C1_planes.npz and the RMSE table to <out>/c1_rmse.txt; then tools/e2e_rmse.sh's 512x512x128 deck with today's kernels, exact and
# native.   usage: tools/reference_session_c1.sh <out dir>
set -u
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R"; O="$1"; mkdir -p "$O"
W=$(mktemp -d)
for tag in ref_fp32 ref_shipped ours_fp32 ours_fp16c_exact ours_fp16c_native; do cp -r tests/golden/refcases/CaseC1 "$W/$tag"; done
( cd oracle/_ref && timeout -k 10 300 ./FluidX3D_fp32 "$W/ref_fp32/conf.luwpf" > "$W/ref_fp32/console.log" 2>&1 </dev/null ); echo "reference fp32 rc=$?"
( cd oracle/_ref && timeout -k 10 300 ./FluidX3D "$W/ref_shipped/conf.luwpf" > "$W/ref_shipped/console.log" 2>&1 </dev/null ); echo "reference shipped rc=$?"
latticeurbanwind_amd/host/luw_driver "$W/ours_fp32/conf.luwpf" --ddf fp32 > "$W/ours_fp32/console.log" 2>&1; echo "driver fp32 rc=$?"
latticeurbanwind_amd/host/luw_driver "$W/ours_fp16c_exact/conf.luwpf" --ddf fp16c --arith exact > "$W/ours_fp16c_exact/console.log" 2>&1; echo "driver fp16c exact rc=$?"
latticeurbanwind_amd/host/luw_driver "$W/ours_fp16c_native/conf.luwpf" --ddf fp16c --arith native > "$W/ours_fp16c_native/console.log" 2>&1; echo "driver fp16c native rc=$?"
for tag in ref_fp32 ref_shipped; do sed 's/\x1b\[[0-9;]*[A-Za-z]//g' "$W/$tag/console.log" | tr '\r' '\n' | grep -v MLUPs | grep -vE '^\|\s+[0-9]+\s+\|' > "$O/$tag.console.txt"; done
python3 - "$W" "$O" <<'PY' | tee "$O/c1_rmse.txt"
import sys, os, glob
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
from vtkio import read_vtk
W, O = sys.argv[1], sys.argv[2]
fac = np.float32(7.838) / np.float32(0.1)              # si_ref_u = max profile U = 7.838 m/s at u_lbm = 0.1
def load(tag, pat, key="data"):
    f = glob.glob(os.path.join(W, tag, "RESULTS", "vtk", pat))[0]
    return read_vtk(f)[1][key]
fields = {}
for tag in ("ref_fp32", "ref_shipped", "ours_fp32", "ours_fp16c_exact", "ours_fp16c_native"):
    fields[tag] = {"u50": load(tag, "*_raw_u-000000050.vtk"), "u100": load(tag, "*_raw_u-000000100.vtk"), "u_avg": load(tag, "*_avg-000000100.vtk", "u_avg")}
solid = load("ref_fp32", "*_avg-000000100.vtk", "fluid")[..., 0] == 0
fluid = ~solid
def rmse(a, b):
    d = ((a - b) / fac)[fluid].astype(np.float64)
    return float(np.sqrt((d ** 2).sum(-1).mean()))
print("CaseC1 (BASELINE configs[0] as a deck): 128^3 cells, %d non-solid; u RMSE in lattice units at K = 50 / K = 100 / u_avg (last 4 steps)" % int(fluid.sum()))
for a, b, what in (("ours_fp32", "ref_fp32", "luw_driver --ddf fp32 against the reference's FP32 build"),
                   ("ours_fp16c_exact", "ref_shipped", "luw_driver --ddf fp16c --arith exact against the shipped build"),
                   ("ours_fp16c_native", "ref_shipped", "luw_driver --ddf fp16c (native, the default) against the shipped build"),
                   ("ours_fp16c_native", "ours_fp16c_exact", "native against exact (this repo, both FP16C)"),
                   ("ref_shipped", "ref_fp32", "the reference's shipped FP16C build against its own FP32 build"),
                   ("ours_fp16c_native", "ref_fp32", "luw_driver --ddf fp16c (native) against the reference's FP32 build")):
    print("  %-78s %.3e  %.3e  %.3e" % (what, rmse(fields[a]["u50"], fields[b]["u50"]), rmse(fields[a]["u100"], fields[b]["u100"]),
        rmse(fields[a]["u_avg"], fields[b]["u_avg"])))
for tag in ("ref_fp32", "ref_shipped"):
    d = {}
    for k, a in fields[tag].items():
        nz, ny, nx, _ = a.shape
        d[k + "_xy"] = a[nz // 2].astype(np.float32); d[k + "_xz"] = a[:, ny // 2].astype(np.float32); d[k + "_yz"] = a[:, :, nx // 2].astype(np.float32)
        d[k + "_stats"] = np.array([a[fluid].min(0), a[fluid].max(0), a[fluid].astype(np.float64).mean(0)], np.float64)    # min / max / mean per component
    d["solid_xy"] = solid[solid.shape[0] // 2]; d["solid_xz"] = solid[:, solid.shape[1] // 2]; d["solid_yz"] = solid[:, :, solid.shape[2] // 2]
    d["dims"] = np.array(fields[tag]["u100"].shape[:3][::-1]); d["solid_count"] = np.array(int(solid.sum()))
    np.savez_compressed(os.path.join(O, tag + "_C1_planes.npz"), **d)
    print("wrote", tag + "_C1_planes.npz", os.path.getsize(os.path.join(O, tag + "_C1_planes.npz")), "bytes")
PY
rm -rf "$W"
echo "== tools/e2e_rmse.sh 100 (512x512x128 deck, one building, nudging + sponge), exact then native FP16C" | tee "$O/e2e_rmse.txt"
LUW_E2E_ARITH=exact tools/e2e_rmse.sh 100 2>&1 | tee -a "$O/e2e_rmse.txt"
LUW_E2E_ARITH=native tools/e2e_rmse.sh 100 2>&1 | grep FP16C | sed 's/$/  [--arith native]/' | tee -a "$O/e2e_rmse.txt"
