for lib in "" $1 $2; do
  d=/tmp/prof_ab_$(basename "${lib:-shipped}" .so); rm -rf $d
  (cd /tmp && TPGAN_HIP_LIBRARY=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-extra > /dev/null 2> $d.err)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== ${lib:-shipped (nt, u2)}"; grep -h "rowcombine_fwd\|rowbn_stats_kernel\|mlp_fwd_kernel<64, 128" $f | awk -F'","' '{gsub(/"/,"",$0); print substr($1,1,70), $2, $4}' | head -6
  grep "timed region" $d.err
done
