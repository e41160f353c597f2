mkdir -p gpurun_out/s4
timeout -k 10 400 python tools/tune.py --rounds 4 l12=leaf_min:12 l14=leaf_min:14 r36=retire_min:36 r44=retire_min:44 s325=path_slots:3407872 s375=path_slots:3932160 l12r36=leaf_min:12,retire_min:36 base2=leaf_min:16 2>&1 | grep -v amdgpu | tee gpurun_out/s4/sweep_final.txt
