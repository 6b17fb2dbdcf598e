# GPU box: does the row order matter for the decomposed step (one block per row in every box)?  rank-shape blocks, LUW_XCD_ROWS = 0 / 1 / 4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z2; mkdir -p $O; : > $O/ab_xcd_rank_blocks.txt
for rep in 1 2 3; do for blk in c4_rank_4x2x1_f32 c5_rank_4x2x1_fp16c_coriolis; do for m in 0 1 4; do
  out=$(LUW_XCD_ROWS=$m timeout -k 10 300 python3 $R/bench.py --rank-shape-block $blk --steps 200 --warmup 20 2>/dev/null | tail -1)
  echo "$blk xcd_rows=$m $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))" 2>&1 | tail -1)" | tee -a $O/ab_xcd_rank_blocks.txt
done; done; done
