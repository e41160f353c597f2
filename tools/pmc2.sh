# usage: bash tools/pmc2.sh <tag> "<counters>" [bench args]  — one counter pass
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_x -- python bench.py --steps 1 --warmup 0 --spp 40 --no-cpu-baseline --no-roofline --no-other-configs "$@" > gpurun_out/pmc_x.log 2>&1 || { tail -5 gpurun_out/pmc_x.log; exit 1; }
python tools/pmc_summary.py gpurun_out/pmc_x > gpurun_out/pmc_${tag}.txt
rm -rf gpurun_out/pmc_x
head -24 gpurun_out/pmc_${tag}.txt
