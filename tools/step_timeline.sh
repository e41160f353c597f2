cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/tl
rm -rf $out
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $root/tools/tune.py --rounds 1 > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
grep '^default' $out.log
f=$(find $out -name '*kernel_trace.csv' | head -1)
python3 $root/tools/step_timeline.py $f
rm -rf $out
