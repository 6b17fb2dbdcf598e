#!/usr/bin/env bash
# First contact with a multi-GPU node (nothing that crosses a device boundary has ever run: one-GPU boxes only).  Stage by stage, a FRESH process per
# stage, the first failure ends the script with that stage's number as exit code; nothing is retried and no process that has touched a GPU execs another
# program (each stage is a child of this shell).
#   1  the node as the HIP runtime sees it: devices, PCI bus ids, the peer-access / link-type / hop matrix (luw_device_info, luw_p2p_info)
#   2  the one-process host across devices: tests/test_gpu_group.py -k distinct_devices (peer stores into another GPU's buffers, cross-device stream waits,
#      hipMemcpyPeerAsync, multi-rank RCCL communicators; one host thread and one per domain)
#   2b the cross-device DEFAULTS by data (tools/xgmi_store_probe.hip + tools/first_contact_defaults.py): what device A's kernels get when they store into
#      device B's memory in the three shapes of the exchange (256-byte rows, one 4-byte / 2-byte element per workgroup); then the one-process host on
#      the literal cut with x faces fused / packed, one round / three phases, peer / staged / rccl, shell-first / whole box, one / many host threads --
#      fresh process each, interleaved, all bit-equal or the stage fails -> $OUT/defaults.json with the winners and the rule applied
#   3  two ranks over RCCL: bench.py --gpus 2 --steps 20 -- its self-check against the CPU oracle through the real transport must pass before anything is timed
#   4  all GPUs: bench.py --gpus N (N = device count, at most 8) -- the line carries rccl.world_size, every rank's PCI bus id and the link type to each
#      halo neighbour, and the one-process host's block
#   5  stage 2 once more under schedule fuzzing (LUW_SCHEDULE_JITTER: random delays of up to 400 us in front of the library's kernels -- an ordering that
#      only held because of how long kernels take on ONE device shows as a difference from the oracle)
#   6  stage 3 once more under schedule fuzzing: the self-check is what counts, the timed value means nothing
# usage: tools/first_contact.sh [--share-device D] [--dry-run] [--out DIR]
#   --share-device D  rehearsal on a ONE-GPU box: stages 3 and 4 with every rank on device D and the faces staged through gloo (stage 4 with 4 ranks: a test
#                     box allows six processes on its GPU); stage 2 is skipped by its own test (needs two GPUs)
#   --dry-run         rehearsal of the control flow WITHOUT a GPU: every stage prints its command instead of running it; FIRST_CONTACT_FAIL=<n> makes stage n fail
set -u
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R"
SHARE=""; DRY=0; OUT="gpurun_out/first_contact"
while [ $# -gt 0 ]; do case "$1" in --share-device) SHARE="$2"; shift 2;; --dry-run) DRY=1; shift;; --out) OUT="$2"; shift 2;; *) echo "unknown option $1" >&2; exit 64;; esac; done
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
stage() { # stage <number> <what> <command...>: runs it as a child process, logs to $OUT/stage<number>.log, ends the script on failure
  local n="$1" what="$2"; shift 2
  local code="$n"; [ "$n" = 2b ] && code=20
  echo "== stage $n: $what"
  if [ "$DRY" = 1 ]; then
    echo "   [dry run] $*" | tee "$OUT/stage$n.log"
    if [ "${FIRST_CONTACT_FAIL:-0}" = "$n" ]; then echo "== stage $n FAILED (injected)"; exit "$code"; fi
    return 0
  fi
  "$@" > "$OUT/stage$n.log" 2>&1; local rc=$?
  tail -n 12 "$OUT/stage$n.log" | sed 's/^/   /'
  if [ $rc -ne 0 ]; then echo "== stage $n FAILED (exit $rc): $OUT/stage$n.log"; exit "$code"; fi
}
NDEV=2
if [ "$DRY" != 1 ]; then NDEV=$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0); fi
PORT=$((29500 + ($$ % 2000)))
stage 1 "devices and links" python3 -c "
import sys; sys.path.insert(0, '$R')
import latticeurbanwind_amd as luw
from latticeurbanwind_amd import capi
import torch
luw.load(); n = torch.cuda.device_count()
for d in range(n): print(capi.device_info(d))
print('from\\\\to ' + ' '.join('%9d' % j for j in range(n)))
for i in range(n):
    row = []
    for j in range(n):
        p = capi.p2p_info(i, j) if i != j else None
        row.append('     self' if p is None else '%9s' % ('%s/%s%s' % (p['link'], p['hops'], '' if p['can_access'] else '!')))
    print('%7d  ' % i + ' '.join(row))
print('(link type / hops; ! = no peer access)')
assert n >= 1
"
stage 2 "one-process host across devices" python3 -m pytest tests/test_gpu_group.py -k distinct_devices -x -q
stage2b() ( # (a subshell) the store probe between device 0 and device 1 (a rehearsal on itself with one device), then the defaults matrix over all devices
  set -e
  [ -x tools/xgmi_store_probe ] || hipcc --offload-arch=gfx950 -O2 -o tools/xgmi_store_probe tools/xgmi_store_probe.hip
  if [ -n "$SHARE" ] || [ "$NDEV" -lt 2 ]; then
    tools/xgmi_store_probe "${SHARE:-0}" "${SHARE:-0}" 65536 5
    python3 tools/first_contact_defaults.py --devices "$(printf "${SHARE:-0},%.0s" 1 2 3 4 5 6 7 8 | sed 's/,$//')" --reps 1 --size 128 32 32 --steps 4 --out "$OUT"
  else
    tools/xgmi_store_probe 0 1; tools/xgmi_store_probe 1 0
    if [ "$NDEV" -ge 8 ]; then python3 tools/first_contact_defaults.py --devices 0,1,2,3,4,5,6,7 --n-gpu 4 2 1 --out "$OUT"
    else python3 tools/first_contact_defaults.py --devices 0,1 --n-gpu 2 1 1 --out "$OUT"; fi
  fi
)
stage 2b "cross-device defaults by data (store shapes; fused / packed, one round / three phases, peer / staged / rccl, shell-first / whole box)" stage2b
if [ -n "$SHARE" ]; then EXTRA=(--share-device "$SHARE" --size 384 64 64); N4=4; else EXTRA=(); N4=$(( NDEV < 8 ? NDEV : 8 )); fi
check_line() { # the printed line of a bench.py --gpus N run: parity passed, the communicator has N ranks, every rank names its bus id and links
  python3 -c "
import json, sys
lines = [l for l in open('$1') if l.startswith('{')]
assert len(lines) == 1, 'expected ONE JSON line, found %d' % len(lines)
d = json.loads(lines[0]); n = int('$2')
assert d['n_gpus'] == n and d['value'] and d['parity'].get('ok') is True and d['rccl']['world_size'] == n, d
assert len(d['ranks']) == n and all(r['bus'] and r['links'] for r in d['ranks']), d['ranks']
print('   %d ranks: %.0f MLUPS, %.3f ms per step, parity ok (%d cases), links %s' % (n, d['value'], d['ms_per_step'], d['parity']['cases'], sorted({v for r in d['ranks'] for v in r['links'].values() if v})))
"
}
# the plain command, as the driver's SCALE runs give it: bench.py starts its own ranks (a child torch.distributed.run, benchmarks/launch.py)
bench_n() { env LUW_BENCH_MASTER_PORT="$2" python3 bench.py --gpus "$1" --steps 20 --warmup 5 "${EXTRA[@]}"; }
stage 3 "two ranks over RCCL, self-check first" bench_n 2 "$PORT"
[ "$DRY" = 1 ] || check_line "$OUT/stage3.log" 2 || { echo "== stage 3 FAILED (line check)"; exit 3; }
[ -n "$SHARE" ] && EXTRA+=(--no-group-host)      # (four ranks + a child of rank 0 + the caller: more processes than a test box allows on its GPU)
stage 4 "$N4 ranks" bench_n "$N4" $((PORT + 1))
[ "$DRY" = 1 ] || check_line "$OUT/stage4.log" "$N4" || { echo "== stage 4 FAILED (line check)"; exit 4; }
stage 5 "one-process host across devices, schedule fuzzing" env LUW_SCHEDULE_JITTER=7:400 python3 -m pytest tests/test_gpu_group.py -k distinct_devices -x -q
[ -n "$SHARE" ] && EXTRA=(--share-device "$SHARE" --size 384 64 64)
stage 6 "two ranks over RCCL, schedule fuzzing" env LUW_SCHEDULE_JITTER=11:400 LUW_BENCH_MASTER_PORT=$((PORT + 2)) python3 bench.py --gpus 2 --steps 20 --warmup 5 \
  --no-secondary "${EXTRA[@]}"
[ "$DRY" = 1 ] || check_line "$OUT/stage6.log" 2 || { echo "== stage 6 FAILED (line check)"; exit 6; }
echo "== first contact complete: $OUT"
