# usage: bash tools/ab_libs.sh <variant .so>...  — A/B of library builds on the headline scene, same box, alternating, one and four streams
keep=/tmp/rrt_keep_$$.so
cp rrt_amd/librrt_amd.so $keep
for rep in 1 2; do
  for v in current "$@"; do
    if [ "$v" = current ]; then cp $keep rrt_amd/librrt_amd.so; else cp "$v" rrt_amd/librrt_amd.so; fi
    for streams in ${AB_STREAMS:-2 4 1}; do
      echo "== $v streams $streams: $(RRT_STREAMS=$streams timeout -k 10 200 python tools/tune.py --rounds 3 2>/dev/null | grep '^default')"
    done
  done
done
cp $keep rrt_amd/librrt_amd.so
