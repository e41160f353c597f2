set -x
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench3.json 2> gpurun_out/bench3.err; echo exit=$?; cat gpurun_out/bench3.json
RRT_BENCH_ONE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --spp 100 --backend gloo > gpurun_out/bench3_n2.json 2> gpurun_out/bench3_n2.err; echo exit=$?; cat gpurun_out/bench3_n2.json; tail -3 gpurun_out/bench3_n2.err
