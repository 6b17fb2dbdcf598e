#!/usr/bin/env bash
# Deck in -> VTK out, wall clock: the real reference (shipped build: FP16C DDFs + thermal lattice, and its FP32 build) against this
# repo's driver on the SAME deck on the GPU box.  The deck is a 512x512x128 profile case with one building, nudging and sponge on,
# NSTEP steps (optionally with the VK inlet) of which the last PURGE are time-averaged (the reference's "mean-field stage": a device->host copy and a host loop
# per sample).  usage (via gpurun): tools/e2e_wall.sh [NSTEP] [PURGE] [vk]      prints one line per run
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"; cd "$R"
NSTEP="${1:-1500}"; PURGE="${2:-500}"; VK="${3:-novk}"      # third argument "vk": von-Karman synthetic-turbulence inlet on (the decks' default)
W=$(mktemp -d)
python3 - "$W" "$NSTEP" "$PURGE" "$VK" <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import make_refcases as mr
mr.write_case(sys.argv[1], "E2E", 1.0, ["enable_buffer_nudging = true", "enable_top_sponge = true", "sponge_thickness_m = 64"], dims=(1024, 1024, 192), building=True,
              nstep=int(sys.argv[2]), unsteady=0, purge=int(sys.argv[3]), vk=(sys.argv[4] == "vk"))
PY
grep -n "n_steps\|purge\|cell_size\|si_x_cfd\|si_y_cfd\|si_z_cfd" "$W/E2E/conf.luwpf" | tr '\n' ' '; echo
run() { # label, binary, options that follow the deck path...
  local label="$1"; local bin="$2"; shift 2
  local c=$(mktemp -d); cp -r "$W/E2E/." "$c/"
  local t0=$(date +%s.%N)
  "$bin" "$c"/conf.luwpf "$@" > "$c/console.log" 2>&1 </dev/null; local rc=$?
  local t1=$(date +%s.%N)
  local grid=$(grep -a -m1 "Grid Resolution" "$c/console.log" | sed 's/\x1b\[[0-9;]*[A-Za-z]//g' | tr -s ' ')
  printf "%-34s rc=%d wall %7.2f s  files %d  %s\n" "$label" "$rc" "$(python3 -c "print($t1 - $t0)")" "$(ls "$c"/RESULTS/vtk/*.vtk 2>/dev/null | wc -l)" "$grid"
  rm -rf "$c"
}
if [ -x oracle/_ref/FluidX3D ]; then
  ( cd oracle/_ref && run "reference shipped (FP16C+T)" ./FluidX3D )
  ( cd oracle/_ref && run "reference FP32 build" ./FluidX3D_fp32 )
fi
run "this repo, --ddf fp16c" "$R/latticeurbanwind_amd/host/luw_driver" --ddf fp16c
run "this repo, --ddf fp32" "$R/latticeurbanwind_amd/host/luw_driver" --ddf fp32
rm -rf "$W"
