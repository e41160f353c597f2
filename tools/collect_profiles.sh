# Collects the rocprofv3 evidence behind bench.py's numbers into gpurun_out/prof_<tag>/ (copy the summaries to profiles/).
# usage: bash tools/collect_profiles.sh <tag>
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $out/bench_under_rocprof.json 2> $out/trace.log || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-configs > /dev/null 2> $out/pmc_f.log || exit 1
python tools/pmc_summary.py $out/pmc_f > $out/pmc_fetch.txt
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_w -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-configs > /dev/null 2> $out/pmc_w.log || exit 1
python tools/pmc_summary.py $out/pmc_w > $out/pmc_write.txt
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $out/pmc_s -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-configs > /dev/null 2> $out/pmc_s.log || exit 1
python tools/pmc_summary.py $out/pmc_s > $out/pmc_sq.txt
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_t -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-configs > /dev/null 2> $out/pmc_t.log || exit 1
python tools/pmc_summary.py $out/pmc_t > $out/pmc_sq2.txt
rm -rf $out/trace $out/pmc_f $out/pmc_w $out/pmc_s $out/pmc_t
ls -la $out; cat $out/kernel_stats.csv | cut -c1-160 | head -8; head -6 $out/pmc_fetch.txt $out/pmc_write.txt
