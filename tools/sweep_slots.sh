# usage: bash tools/sweep_slots.sh  — path-pool size x number of pool parts (streams) on the headline scene
for streams in 4 2 1; do
  echo "== RRT_STREAMS $streams"
  RRT_STREAMS=$streams timeout -k 10 200 python tools/tune.py --rounds 2 s512k=path_slots:524288 s1m=path_slots:1048576 s2m=path_slots:2097152 s3m=path_slots:3145728 s4m=path_slots:4194304 s6m=path_slots:6291456 | grep -v amdgpu || exit 1
done
