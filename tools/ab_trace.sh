# usage: bash tools/ab_trace.sh <variant .so>...  — per-kernel time totals of library builds on the headline scene, two streams, under the kernel trace
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
keep=/tmp/rrt_keep_$$.so
cp $root/rrt_amd/librrt_amd.so $keep
for v in current "$@"; do
  if [ "$v" = current ]; then cp $keep $root/rrt_amd/librrt_amd.so; else cp "$root/$v" $root/rrt_amd/librrt_amd.so; fi
  tag=$(basename "$v" .so)
  out=$root/gpurun_out/abtrace_$tag
  rm -rf $out
  RRT_STREAMS=${AB_STREAMS:-2} timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $root/tools/tune.py --rounds 3 > $out.log 2>&1 || { echo "trace of $v failed"; cp $keep $root/rrt_amd/librrt_amd.so; exit 1; }
  echo "== $v"; grep '^default' $out.log
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:10.2f} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
done
cp $keep $root/rrt_amd/librrt_amd.so
