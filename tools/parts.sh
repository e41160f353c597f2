# usage: bash tools/parts.sh  — headline scene with the path pool in 1..4 parts (RRT_STREAMS)
for k in 1 2 3 4; do echo "== parts $k"; RRT_STREAMS=$k timeout -k 10 200 python tools/tune.py --rounds 2 "$@" || exit 1; done
