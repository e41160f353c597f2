# usage: bash tools/variants.sh <variant .so>...  — times each library variant on the headline scene (one stream)
for v in default "$@"; do
  if [ "$v" != default ]; then cp rrt_amd/librrt_amd.so /tmp/rrt_keep.so && cp "$v" rrt_amd/librrt_amd.so; fi
  echo "== $v"; RRT_STREAMS=1 timeout -k 10 200 python tools/tune.py --rounds 2 || exit 1
  if [ "$v" != default ]; then cp /tmp/rrt_keep.so rrt_amd/librrt_amd.so; fi
done
