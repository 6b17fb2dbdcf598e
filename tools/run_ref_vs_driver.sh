#!/usr/bin/env bash
# usage: run_ref_vs_driver.sh CaseX [CaseY ...]: real reference (FP32 + shipped builds) and this repo's driver on the same decks,
# outputs under gpurun_out/ref/<build>_<case> and gpurun_out/mine/<ddf>_<case> (to be packed by tests/golden/pack_ref_outputs.py)
set -u
for c in "$@"; do
  oracle/run_ref_case.sh oracle/_ref/FluidX3D_fp32 tests/golden/refcases/$c gpurun_out/ref/fp32_$c 240 >/dev/null
  oracle/run_ref_case.sh oracle/_ref/FluidX3D tests/golden/refcases/$c gpurun_out/ref/shipped_$c 240 >/dev/null
  for ddf in fp32 fp16c; do
    w=$(mktemp -d); cp -r tests/golden/refcases/$c/. $w/
    latticeurbanwind_amd/host/luw_driver $w/conf.luw* --ddf $ddf > $w/console.log 2>&1; echo "driver $c $ddf rc=$?"
    mkdir -p gpurun_out/mine/${ddf}_$c; cp $w/RESULTS/vtk/*.vtk $w/RESULTS/*.csv $w/console.log gpurun_out/mine/${ddf}_$c/ 2>/dev/null
  done
done
