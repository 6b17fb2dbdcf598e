#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace + HBM / SQ counter passes (tools/profile_bench.sh) for the headline and every secondary single-GPU block of bench.py, then
# the summaries (tools/summarize_profile.py) -> profiles/<round>_<key>_summary.json + _kernel_stats.csv (copied to gpurun_out/profiles_out/ for the merge back).
# usage: profile_all.sh <round tag, e.g. r03> [block ...]   (default: all)
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
RT="$1"; shift
ALL="c3_f32 c3_fp16c c3_fp16c_cor c3_fp16c_th c2_f32 c2_fp16c cube_f32 cube_fp16c urban_fp16c_cor"
for blk in ${@:-$ALL}; do
  case $blk in
    c3_f32) key=f32_1024x1024x256_bld; args="--workload c3"; kern=k_stream_collide_s; algo="";;
    # FP16C: the bit-exact kernels (--arith exact) under the plain keys, the native-arithmetic ones (bench.py's and the driver's default) under _nat
    c3_fp16c) key=fp16c_1024x1024x256_bld; args="--workload c3 --dtype fp16c --arith exact"; kern=k_stream_collide_p; algo="";;
    c3_fp16c_cor) key=fp16c_1024x1024x256_bld_cor; args="--workload c3 --dtype fp16c --coriolis --arith exact"; kern=k_stream_collide_p; algo="";;
    c3_fp16c_th) key=fp16c_1024x1024x256_bld_th; args="--workload c3 --dtype fp16c --thermal --arith exact"; kern=k_stream_collide_p; algo="";;
    c2_f32) key=f32_512x512x512; args="--workload c2"; kern=k_stream_collide_s; algo="";;
    c2_fp16c) key=fp16c_512x512x512; args="--workload c2 --dtype fp16c --arith exact"; kern=k_stream_collide_p; algo="";;
    c2_fp16c_nat) key=fp16c_512x512x512_nat; args="--workload c2 --dtype fp16c --arith native"; kern=k_stream_collide_p; algo="";;
    cube_f32) key=f32_1024x1024x1024; args="--workload cube1024"; kern=k_stream_collide_s; algo="";;
    cube_fp16c) key=fp16c_1024x1024x1024; args="--workload cube1024 --dtype fp16c --arith exact"; kern=k_stream_collide_p; algo="";;
    cube_fp16c_nat) key=fp16c_1024x1024x1024_nat; args="--workload cube1024 --dtype fp16c --arith native"; kern=k_stream_collide_p; algo="";;
    urban_fp16c_cor) key=fp16c_512x512x512_bld_urban_cor; args="--workload tile512 --urban --dtype fp16c --coriolis --arith exact"; kern=k_stream_collide_p; algo="";;
    # the same FP16C blocks with the native-arithmetic kernels (LUW_OPT_NATIVE_ARITH)
    c3_fp16c_nat) key=fp16c_1024x1024x256_bld_nat; args="--workload c3 --dtype fp16c --arith native"; kern=k_stream_collide_p; algo="";;
    c3_fp16c_cor_nat) key=fp16c_1024x1024x256_bld_cor_nat; args="--workload c3 --dtype fp16c --coriolis --arith native"; kern=k_stream_collide_p; algo="";;
    c3_fp16c_th_nat) key=fp16c_1024x1024x256_bld_th_nat; args="--workload c3 --dtype fp16c --thermal --arith native"; kern=k_stream_collide_p; algo="";;
    urban_fp16c_cor_nat) key=fp16c_512x512x512_bld_urban_cor_nat; args="--workload tile512 --urban --dtype fp16c --coriolis --arith native"; kern=k_stream_collide_p; algo="";;
  esac
  "$R/tools/profile_bench.sh" "${RT}_$key" --steps 40 --warmup 8 $args > /dev/null 2>&1
  python3 "$R/tools/summarize_profile.py" "${RT}_$key" $kern $algo | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['tag'], 'rocprof', d['rocprof_avg_ms'], 'events', d['hip_event_avg_ms_same_run'], 'traffic/algo', d['traffic_over_algorithmic'], 'VALUBusy', d['VALUBusy_percent'], 'valu/wave', d['valu_insts_per_wave'])"
  mkdir -p "$R/gpurun_out/profiles_out"; cp "$R/profiles/${RT}_${key}_summary.json" "$R/profiles/${RT}_${key}_kernel_stats.csv" "$R/gpurun_out/profiles_out/" 2>/dev/null
done
