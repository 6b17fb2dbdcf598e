python -m pytest tests/test_gpu_parity.py -q -m gpu -k "codec or matches_oracle" 2>&1 | tail -2
for r in 1 2; do for lib in "" /root/repo/gpurun_noslp.so; do for dt in f32 fp16c; do
  echo -n "lib=${lib:-default} $dt: "; LUW_CORE_LIB=$lib python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --dtype $dt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done; done; done
