# usage: bash tools/wave_timing.sh <library built with -DRRT_WAVE_TIMING>  — wave-slot utilisation inside extend and shade, one and two streams
keep=/tmp/rrt_keep_$$.so
cp rrt_amd/librrt_amd.so $keep
cp "$1" rrt_amd/librrt_amd.so
for streams in ${WT_STREAMS:-1}; do
  echo "== streams $streams"
  RRT_STREAMS=$streams timeout -k 10 200 python tools/wave_timing.py 2>&1 | tail -6
done
cp $keep rrt_amd/librrt_amd.so
