R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05v; mkdir -p $O; : > $O/ab_xcd_g_headline.txt
for rep in 1 2 3 4; do for blk in headline c2_f32 tile512_f32; do for m in 0 2 4; do
  case $blk in headline) args="";; c2_f32) args="--workload c2";; tile512_f32) args="--workload tile512 --urban";; esac
  out=$(LUW_XCD_ROWS=$m timeout -k 10 300 python3 $R/bench.py $args --no-secondary --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1)
  echo "$blk xcd_rows=$m $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))" 2>&1 | tail -1)" | tee -a $O/ab_xcd_g_headline.txt
done; done; done
