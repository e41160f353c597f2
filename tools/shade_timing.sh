# usage: bash tools/shade_timing.sh <library built with -DRRT_SHADE_TIMING>  — cycles per stage of the shade kernel, one stream
keep=/tmp/rrt_keep_$$.so
cp rrt_amd/librrt_amd.so $keep
cp "$1" rrt_amd/librrt_amd.so
RRT_STREAMS=${ST_STREAMS:-1} SPP=${SPP:-100} timeout -k 10 200 python tools/shade_timing.py 2>&1 | tail -3
cp $keep rrt_amd/librrt_amd.so
