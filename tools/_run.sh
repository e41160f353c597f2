set -x
bash tools/collect_profiles.sh r03_e > gpurun_out/collect_r03_e.log 2>&1 || { tail -5 gpurun_out/collect_r03_e.log; exit 1; }
bash tools/collect_single_stream.sh r03_e > gpurun_out/collect_r03_e_1s.log 2>&1 || { tail -5 gpurun_out/collect_r03_e_1s.log; exit 1; }
timeout -k 10 200 python tools/extend_passes.py > gpurun_out/prof_r03_e/extend_passes.txt 2>&1 || exit 1
rm -rf gpurun_out/prof_cfg3; bash tools/cfg3_profile.sh > gpurun_out/collect_cfg3.log 2>&1 || { tail -5 gpurun_out/collect_cfg3.log; exit 1; }
timeout -k 10 300 python bench.py > gpurun_out/prof_r03_e/bench.json 2> gpurun_out/prof_r03_e/bench.err || exit 1
bash tools/configs.sh > gpurun_out/prof_r03_e/configs.txt 2>&1
cat gpurun_out/prof_r03_e/configs.txt
