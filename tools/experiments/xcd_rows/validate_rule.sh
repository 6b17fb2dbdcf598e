# GPU box: luw_create's rule (LUW_XCD_ROWS unset) against the dispatch order (LUW_XCD_ROWS=0), the shapes the rule switches on + bench blocks
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05w; mkdir -p $O; : > $O/xcd_rule_validation.txt
run() { # label, env, args
  out=$(env $2 timeout -k 10 300 python3 $R/bench.py $3 --steps 100 --warmup 10 2>/dev/null | tail -1)
  echo "$1 [$2] $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))" 2>&1 | tail -1)" | sed 's/\[X=1\]/[rule]/; s/\[LUW_XCD_ROWS=0\]/[dispatch order]/' | tee -a $O/xcd_rule_validation.txt
}
for rep in 1 2; do
  for shp in "1024 1024 512" "1024 1024 1024" "2048 1024 512" "2048 2048 256"; do for e in LUW_XCD_ROWS=0 X=1; do
    run "f32 $shp" $e "--workload c2 --size $shp --no-secondary --no-cpu-baseline"
  done; done
  for shp in "1024 1024 1024" "2048 1024 512"; do for e in LUW_XCD_ROWS=0 X=1; do
    run "fp16c $shp" $e "--workload c2 --size $shp --dtype fp16c --no-secondary --no-cpu-baseline"
  done; done
  for e in LUW_XCD_ROWS=0 X=1; do run "headline" $e "--no-secondary --no-cpu-baseline"; run "c3_fp16c_thermal" $e "--secondary-block c3_fp16c_thermal"; done
done
