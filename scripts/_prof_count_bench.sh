# counters of the two counting kernels in the standalone scoring-chain benchmark (scripts/count_kernel_bench.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for P in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"; do
  D=gpurun_out/r05_cb_pmc
  rocprofv3 --pmc $P -d $D -o q --output-format csv -- python3 scripts/count_kernel_bench.py > $D.log 2>&1
  python3 scripts/pmc_kernel.py $D ransac_count_mfma_kernel
  python3 scripts/pmc_kernel.py $D ransac_count_kernel
  rm -rf $D
done
