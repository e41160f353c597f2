timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1; echo exit=$?; tail -4 gpurun_out/t.log
echo "--- one stream"; RRT_STREAMS=1 timeout -k 10 300 python tools/tune.py --rounds 2 "$@"
