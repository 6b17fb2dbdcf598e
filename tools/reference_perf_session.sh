#!/usr/bin/env bash
# GPU box, oracle/_ref/ travelling (tests/golden/README.md): what the REAL reference's own OpenCL kernels do on this MI355X, next to this repo's driver, today.
# (1) the 1024x1024x256 empty-channel deck of make_refcases.py --perf, 200 steps: the reference's own benchmark line ("normal Steps/s") for both builds, and
#     the driver's MLUPs for --ddf fp32 / fp16c;  (2) tools/e2e_wall.sh: deck in -> VTK out, 1500 steps / 500 averaged, and 4000 / 1500 with the VK inlet.
# usage: tools/reference_perf_session.sh <out dir>
set -u
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R"; O="$1"; mkdir -p "$O"
W=$(mktemp -d)
python3 tests/golden/make_refcases.py "$W" --perf > /dev/null
cells=$((1024*1024*256))
{
echo "== reference kernels on this GPU (deck Perf1024x1024x256: 1024x1024x256 cells, empty channel, 200 steps), MLUPS from its own 'normal Steps/s' line"
for b in FluidX3D_fp32 FluidX3D; do
  c=$(mktemp -d); cp -r "$W/Perf1024x1024x256/." "$c/"
  ( cd oracle/_ref && timeout -k 10 400 ./$b "$c/conf.luwpf" > "$c/console.log" 2>&1 </dev/null )
  sps=$(sed 's/\x1b\[[0-9;]*[A-Za-z]//g' "$c/console.log" | tr '\r' '\n' | grep -a "normal Steps/s" | head -1 | sed 's/.*normal Steps\/s = \([0-9.]*\).*/\1/')
  echo "  $b: normal Steps/s = $sps -> $(python3 -c "print('%.0f MLUPS' % ($sps * $cells / 1e6))")"
  rm -rf "$c"
done
for ddf in fp32 fp16c; do
  c=$(mktemp -d); cp -r "$W/Perf1024x1024x256/." "$c/"
  t0=$(date +%s.%N); latticeurbanwind_amd/host/luw_driver "$c/conf.luwpf" --ddf $ddf > "$c/console.log" 2>&1; t1=$(date +%s.%N)
  echo "  luw_driver --ddf $ddf: whole process $(python3 -c "print('%.2f s' % ($t1 - $t0))"); $(grep -a -i "mlups\|steps/s" "$c/console.log" | tail -2 | tr -s ' ' | tr '\n' ';')"
  rm -rf "$c"
done
echo "== deck in -> VTK out (tools/e2e_wall.sh 1500 500)"; tools/e2e_wall.sh 1500 500
echo "== deck in -> VTK out with the von-Karman inlet (tools/e2e_wall.sh 4000 1500 vk)"; tools/e2e_wall.sh 4000 1500 vk
} 2>&1 | tee "$O/reference_perf.txt"
rm -rf "$W"
