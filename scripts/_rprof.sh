# RANSAC timeline + SQ counters of the counting kernel on config 5 (scratch helper)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python3 bench.py --config sfm3 --steps 1 --warmup 1"
rocprofv3 --kernel-trace -d gpurun_out/rtl -o t --output-format csv -- $B > gpurun_out/rtl.log 2>&1 && python3 scripts/ransac_timeline.py gpurun_out/rtl > gpurun_out/rtl_summary.txt
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/rp1 -o q --output-format csv -- $B > gpurun_out/rp1.log 2>&1 && python3 scripts/pmc_kernel.py gpurun_out/rp1 ransac_count_kernel > gpurun_out/rp_summary.txt
rm -rf gpurun_out/rp1 gpurun_out/rtl
cat gpurun_out/rtl_summary.txt gpurun_out/rp_summary.txt
