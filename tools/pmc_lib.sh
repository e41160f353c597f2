# usage: bash tools/pmc_lib.sh <tag> <lib.so|current> "<counters>"  — one counter pass over a 40-spp headline render with the given library build, one stream
tag=$1; libv=$2; ctrs=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
keep=/tmp/rrt_keep_pmc.so
cp rrt_amd/librrt_amd.so $keep
if [ "$libv" != current ]; then cp "$libv" rrt_amd/librrt_amd.so; fi
export RRT_STREAMS=1
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_x -- python bench.py --steps 1 --warmup 0 --spp 40 --no-cpu-baseline --no-roofline --no-other-configs > gpurun_out/pmc_x.log 2>&1
rc=$?
cp $keep rrt_amd/librrt_amd.so
[ $rc -eq 0 ] || { tail -5 gpurun_out/pmc_x.log; exit 1; }
python tools/pmc_summary.py gpurun_out/pmc_x > gpurun_out/pmc_${tag}.txt
rm -rf gpurun_out/pmc_x
head -12 gpurun_out/pmc_${tag}.txt
