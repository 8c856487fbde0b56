cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc10
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc10 -o a --output-format csv -- python3 scripts/kbench.py --frames 410 --reps 1 > gpurun_out/pmc10/loga.txt 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY -d gpurun_out/pmc10 -o b --output-format csv -- python3 scripts/kbench.py --frames 410 --reps 1 > gpurun_out/pmc10/logb.txt 2>&1
echo rc=$?
