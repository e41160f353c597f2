# book2chap2 1200x800x500 (moving spheres): kernel trace + one-stream SQ counters, summaries in gpurun_out/prof_cfg3/
out=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="tools/tune.py --scene 2 --nx 1200 --ny 800 --spp 500 --rounds 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $args > $out/run.txt 2> $out/trace.log || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
export RRT_STREAMS=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace1 -- python $args > $out/run_1stream.txt 2> $out/trace1.log || exit 1
cp $(find $out/trace1 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_1stream.csv
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $out/pmc_s -- python $args > /dev/null 2> $out/pmc_s.log || exit 1
python tools/pmc_summary.py $out/pmc_s > $out/pmc_sq_1stream.txt
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_t -- python $args > /dev/null 2> $out/pmc_t.log || exit 1
python tools/pmc_summary.py $out/pmc_t > $out/pmc_sq2_1stream.txt
rm -rf $out/trace $out/trace1 $out/pmc_s $out/pmc_t
cat $out/run.txt $out/run_1stream.txt; cut -c1-70,150-260 $out/kernel_stats_1stream.csv | head -5
