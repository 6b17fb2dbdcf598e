#!/usr/bin/env bash
# CPU container: the oracle (oracle/luw_oracle.c, both paths) under AddressSanitizer through the tests that drive it hardest -- the row-wise path against the
# literal one, the known answers, the fixtures of the real reference, the gloo worlds of 2-8 ranks with the halo maps and edge messages.  (GPU AddressSanitizer
# is not available on the pool; this is the CPU build only.)  Then the deck driver's host stage (deck parser, setup mathematics, boundary builders, mesh and
# CSV readers) under AddressSanitizer + UndefinedBehaviorSanitizer (non-recovering) through tests/test_driver_host.py and test_driver_robustness.py.
# Restores the product builds afterwards.   usage: tools/cpu_sanitizers.sh
set -uo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
H="$R/latticeurbanwind_amd/host"
# whatever ends this script (a failing command under set -u, a signal, Ctrl-C): the product builds come back
# (a backup copy where there is one, a rebuild from the sources otherwise; the copies live in a directory of this run's own)
B="$(mktemp -d)"
SAN_ORACLE=0; SAN_DRIVER=0
restore_oracle() { if [ -f "$B/libluw_oracle.so" ]; then cp "$B/libluw_oracle.so" "$R/oracle/libluw_oracle.so"; else make -C "$R/oracle" -s clean all; fi; touch "$R/oracle/libluw_oracle.so"; SAN_ORACLE=0; }
restore_driver() { if [ -f "$B/luw_driver" ]; then cp "$B/luw_driver" "$H/luw_driver"; else make -C "$H" -s clean all; fi; touch "$H/luw_driver"; SAN_DRIVER=0; }
restore() {
  [ "$SAN_ORACLE" = 1 ] && restore_oracle
  [ "$SAN_DRIVER" = 1 ] && restore_driver
  rm -rf "$B"
}
trap restore EXIT
cp "$R/oracle/libluw_oracle.so" "$B/libluw_oracle.so" 2>/dev/null
SAN_ORACLE=1
gcc -O1 -g -march=x86-64-v3 -fPIC -std=gnu11 -D_GNU_SOURCE -ffp-contract=off -fno-fast-math -fno-math-errno -fopenmp -fsanitize=address -fno-omit-frame-pointer \
  -shared -o "$R/oracle/libluw_oracle.so" "$R/oracle/luw_oracle.c" -lm && touch "$R/oracle/libluw_oracle.so"
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so)" python3 -m pytest "$R/tests/test_oracle_fast_path.py" "$R/tests/test_oracle_known_answers.py" \
  "$R/tests/test_oracle_vs_reference.py" "$R/tests/test_vk_inlet.py" "$R/tests/test_distributed_gloo.py" "$R/tests/test_bench_selfcheck.py" -q -m "not gpu"
rc=$?
restore_oracle
cp "$H/luw_driver" "$B/luw_driver" 2>/dev/null
SAN_DRIVER=1
( cd "$H" && g++ -std=c++17 -O1 -g -ffp-contract=off -pthread -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Wall \
  -Wno-misleading-indentation -o luw_driver luw_driver.cpp -L../csrc -lluw_core -Wl,-rpath,'$ORIGIN/../csrc' -Wl,-rpath,/opt/rocm/lib && touch luw_driver )
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 python3 -m pytest "$R/tests/test_driver_host.py" "$R/tests/test_driver_robustness.py" -q -m "not gpu"
rc2=$?
restore_driver
exit $((rc | rc2))
