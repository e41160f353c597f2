# A/B of the pool order (camera rays last) against slot order (RRT_FRESH_APART=0): GPU tests, pass statistics, step times of three scenes
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ff_t.log 2>&1; rc=$?; tail -3 gpurun_out/ff_t.log; [ $rc -eq 0 ] || exit $rc
for v in 0 1; do echo "=== RRT_FRESH_APART=$v"; RRT_FRESH_APART=$v timeout -k 10 120 python tools/extend_passes.py 2>&1 | grep -v amdgpu.ids || exit 1; done
echo "=== chap12, two streams"; timeout -k 10 300 python tools/tune.py --rounds 6 off=env_RRT_FRESH_APART:0 on=env_RRT_FRESH_APART:1 2>&1 | grep -v amdgpu.ids || exit 1
echo "=== chap12, one stream, 4 Mi slots"; RRT_STREAMS=1 timeout -k 10 300 python tools/tune.py --rounds 3 off=env_RRT_FRESH_APART:0,path_slots:4194304 on=env_RRT_FRESH_APART:1,path_slots:4194304 2>&1 | grep -v amdgpu.ids || exit 1
echo "=== book2chap2"; timeout -k 10 300 python tools/tune.py --scene 2 --rounds 4 off=env_RRT_FRESH_APART:0 on=env_RRT_FRESH_APART:1 2>&1 | grep -v amdgpu.ids || exit 1
echo "=== 1 M spheres"; timeout -k 10 400 python tools/tune.py --scene 3 --nx 1920 --ny 1080 --spp 256 --half 500 --rounds 3 off=env_RRT_FRESH_APART:0 on=env_RRT_FRESH_APART:1 2>&1 | grep -v amdgpu.ids || exit 1
