# SQ counters of config 5's kernels (RANSAC LM / counting kernels in particular):
#   gpurun -- 'bash scripts/_prof_sfm_pmc.sh r05'   -> gpurun_out/<TAG>_sfm3_pmc.json (+ .txt)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05}
B="python3 bench.py --config sfm3 --steps 1 --warmup 1"
D=gpurun_out/${TAG}_sfm_pmcA
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $D -o q --output-format csv -- $B > $D.log 2>&1
D2=gpurun_out/${TAG}_sfm_pmcB
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 -d $D2 -o q --output-format csv -- $B > $D2.log 2>&1
python3 scripts/collect_sfm_pmc.py $D $D2 gpurun_out/${TAG}_sfm3_pmc.json > gpurun_out/${TAG}_sfm3_pmc.txt 2>&1
rm -rf $D $D2
cat gpurun_out/${TAG}_sfm3_pmc.txt
