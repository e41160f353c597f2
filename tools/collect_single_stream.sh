# Per-kernel counters with the whole path pool on ONE stream (RRT_STREAMS=1): launches do not overlap, so a kernel's
# GRBM_GUI_ACTIVE is its own time and busy fractions can be read per kernel.  usage: bash tools/collect_single_stream.sh <tag>
tag=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_1s
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RRT_STREAMS=1
args="bench.py --steps 1 --warmup 0 --slots 4194304 --no-cpu-baseline --no-roofline --no-other-configs"  # 4 Mi slots: two full generations of extend workgroups on one stream
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $out/pmc_s -- python $args > /dev/null 2> $out/pmc_s.log || exit 1
python tools/pmc_summary.py $out/pmc_s > $out/pmc_sq.txt
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_t -- python $args > /dev/null 2> $out/pmc_t.log || exit 1
python tools/pmc_summary.py $out/pmc_t > $out/pmc_sq2.txt
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -- python $args > /dev/null 2> $out/pmc_f.log || exit 1
python tools/pmc_summary.py $out/pmc_f > $out/pmc_fetch.txt
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_w -- python $args > /dev/null 2> $out/pmc_w.log || exit 1
python tools/pmc_summary.py $out/pmc_w > $out/pmc_write.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $args > $out/bench_under_rocprof.json 2> $out/trace.log || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/pmc_s $out/pmc_t $out/pmc_f $out/pmc_w $out/trace
head -10 $out/pmc_sq.txt $out/pmc_sq2.txt
