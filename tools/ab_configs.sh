keep=/tmp/rrt_keep_$$.so
cp rrt_amd/librrt_amd.so $keep
for v in current "$@"; do
  if [ "$v" = current ]; then cp $keep rrt_amd/librrt_amd.so; else cp "$v" rrt_amd/librrt_amd.so; fi
  echo "== $v"; timeout -k 10 300 bash tools/configs.sh || exit 1
done
cp $keep rrt_amd/librrt_amd.so
