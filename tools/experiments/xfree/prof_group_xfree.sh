cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05m; mkdir -p $O
for mode in no_x_slabs x_slabs; do
  if [ $mode = x_slabs ]; then unset LUW_GROUP_X_SLABS; else export LUW_GROUP_X_SLABS=0; fi
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$mode -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_group.py f32 quick > $O/prof_$mode.log 2>&1 || exit 1
  f=$(find $O/prof_$mode -name "*kernel_stats.csv" | head -1); cp $f $O/group_f32_${mode}_kernel_stats.csv
  find $O/prof_$mode -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O/prof_$mode -name "*.db" -delete
  head -12 $O/group_f32_${mode}_kernel_stats.csv | cut -c1-260
done
