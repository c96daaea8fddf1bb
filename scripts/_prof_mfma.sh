cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
D=gpurun_out/r05_mf_pmc
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $D -o q --output-format csv -- python3 bench.py --config sfm3 --steps 1 --warmup 1 > $D.log 2>&1
python3 scripts/pmc_kernel.py $D ransac_count_mfma_kernel
rm -rf $D
