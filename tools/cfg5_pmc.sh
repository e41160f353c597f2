cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_cfg5b
args="tools/tune.py --scene 3 --half 500 --nx 1920 --ny 1080 --spp 16 --rounds 1"
RRT_STREAMS=1 timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out -- python $args > /dev/null 2> $out.log || exit 1
python tools/pmc_summary.py $out | head -16
rm -rf $out
