# Profiles the 1 M-sphere configuration (HBM BVH walk) at reduced spp: kernel trace + HBM traffic, summaries in gpurun_out/prof_cfg5/
out=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg5
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="tools/tune.py --scene 3 --half 500 --nx 1920 --ny 1080 --spp 32 --rounds 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $args > $out/run.txt 2> $out/trace.log || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -- python $args > /dev/null 2> $out/pmc_f.log || exit 1
python tools/pmc_summary.py $out/pmc_f > $out/pmc_fetch.txt
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_w -- python $args > /dev/null 2> $out/pmc_w.log || exit 1
python tools/pmc_summary.py $out/pmc_w > $out/pmc_write.txt
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $out/pmc_s -- python $args > /dev/null 2> $out/pmc_s.log || exit 1
python tools/pmc_summary.py $out/pmc_s > $out/pmc_sq.txt
rm -rf $out/trace $out/pmc_f $out/pmc_w $out/pmc_s
cat $out/run.txt; cut -c1-60,150-260 $out/kernel_stats.csv | head -6; head -8 $out/pmc_fetch.txt; head -8 $out/pmc_write.txt; head -12 $out/pmc_sq.txt
