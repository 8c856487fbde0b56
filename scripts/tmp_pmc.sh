cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc11
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc11 -o a --output-format csv -- python3 scripts/kbench.py --frames 410 --reps 1 > gpurun_out/pmc11/loga.txt 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY -d gpurun_out/pmc11 -o b --output-format csv -- python3 scripts/kbench.py --frames 410 --reps 1 > gpurun_out/pmc11/logb.txt 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_TA_BUSY_sum -d gpurun_out/pmc11 -o c --output-format csv -- python3 scripts/kbench.py --frames 410 --reps 1 > gpurun_out/pmc11/logc.txt 2>&1
echo rc=$?
