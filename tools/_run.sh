set -x
bash tools/collect_profiles.sh r03_d > gpurun_out/collect_r03_d.log 2>&1 || { tail -5 gpurun_out/collect_r03_d.log; exit 1; }
bash tools/collect_single_stream.sh r03_d > gpurun_out/collect_r03_d_1s.log 2>&1 || { tail -5 gpurun_out/collect_r03_d_1s.log; exit 1; }
timeout -k 10 200 python tools/extend_passes.py > gpurun_out/prof_r03_d/extend_passes.txt 2>&1 || exit 1
rm -rf gpurun_out/prof_cfg3; bash tools/cfg3_profile.sh > gpurun_out/collect_cfg3.log 2>&1 || { tail -5 gpurun_out/collect_cfg3.log; exit 1; }
rm -rf gpurun_out/prof_cfg5; bash tools/cfg5_profile.sh > gpurun_out/collect_cfg5.log 2>&1 || { tail -5 gpurun_out/collect_cfg5.log; exit 1; }
timeout -k 10 200 python tools/extend_passes.py 3 1920 1080 8 500 > gpurun_out/prof_cfg5/extend_passes.txt 2>&1 || exit 1
timeout -k 10 300 python bench.py > gpurun_out/prof_r03_d/bench.json 2> gpurun_out/prof_r03_d/bench.err || exit 1
timeout -k 10 300 python tools/configs.sh > /dev/null 2>&1; bash tools/configs.sh > gpurun_out/prof_r03_d/configs.txt 2>&1
cat gpurun_out/prof_r03_d/configs.txt
