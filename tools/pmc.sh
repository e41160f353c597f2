# usage: bash tools/pmc.sh <tag> [bench args...]   — two SQ counter passes over a short bench run, summaries in gpurun_out/pmc_<tag>_{1,2}.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_a -- python bench.py --steps 1 --warmup 0 --spp 40 --no-cpu-baseline --no-roofline --no-other-configs "$@" > gpurun_out/pmc_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_b -- python bench.py --steps 1 --warmup 0 --spp 40 --no-cpu-baseline --no-roofline --no-other-configs "$@" > gpurun_out/pmc_b.log 2>&1 || exit 1
python tools/pmc_summary.py gpurun_out/pmc_a > gpurun_out/pmc_${tag}_1.txt
python tools/pmc_summary.py gpurun_out/pmc_b > gpurun_out/pmc_${tag}_2.txt
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
head -20 gpurun_out/pmc_${tag}_1.txt; head -20 gpurun_out/pmc_${tag}_2.txt
