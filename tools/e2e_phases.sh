#!/usr/bin/env bash
# Where a deck-in -> VTK-out run of this repo's driver spends its wall time (LUW_DRIVER_TIMING=1), on the deck of tools/e2e_wall.sh.
# usage (via gpurun): tools/e2e_phases.sh [NSTEP] [PURGE] [fp16c|fp32]
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"; cd "$R"
NSTEP="${1:-4000}"; PURGE="${2:-1500}"; DDF="${3:-fp16c}"
W=$(mktemp -d)
python3 - "$W" "$NSTEP" "$PURGE" <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import make_refcases as mr
mr.write_case(sys.argv[1], "E2E", 1.0, ["enable_buffer_nudging = true", "enable_top_sponge = true", "sponge_thickness_m = 64"], dims=(1024, 1024, 192), building=True,
              nstep=int(sys.argv[2]), unsteady=0, purge=int(sys.argv[3]), vk=True)
PY
t0=$(date +%s.%N)
LUW_DRIVER_TIMING=1 "$R/latticeurbanwind_amd/host/luw_driver" "$W/E2E/conf.luwpf" --ddf "$DDF" 2>&1 >/dev/null </dev/null | grep "^\[timing\]"
t1=$(date +%s.%N)
python3 -c "print('[timing] whole process              %8.3f s' % ($t1 - $t0))"
ls -la "$W"/E2E/RESULTS/vtk/ | awk '{print $5, $9}' | tail -3
rm -rf "$W"
