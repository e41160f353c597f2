# usage: bash tools/sweep_slots.sh  — path-pool size x number of pool parts (streams) on the headline scene
for streams in 4 3 2 1; do
  echo "== RRT_STREAMS $streams"
  RRT_STREAMS=$streams timeout -k 10 200 python tools/tune.py --rounds 2 s2m=path_slots:2097152 s3m=path_slots:3145728 s4m=path_slots:4194304 s5m=path_slots:5242880 s6m=path_slots:6291456 s8m=path_slots:8388608 | grep -v amdgpu || exit 1
done
