"""What bench.py measures with: workloads and the CPU baseline (common.py), the N > 1 line (multi.py).  bench.py at the repo root is the entry point."""
