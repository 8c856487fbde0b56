cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc9
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc9 -o a --output-format csv -- python3 scripts/kbench.py --frames 410 --reps 1 > gpurun_out/pmc9/loga.txt 2>&1 &&
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d gpurun_out/pmc9 -o k --output-format csv -- python3 scripts/kbench.py --frames 410 --reps 1 > gpurun_out/pmc9/logk.txt 2>&1
echo rc=$?
