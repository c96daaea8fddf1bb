# SQ / SQC counter passes of the box kernels on the rectified and on tilted 4096^2 pairs (one pyramid run per pass):
#   gpurun -- 'bash scripts/_prof_tilt.sh r05 "0 3 45"'
# writes gpurun_out/<TAG>_pmc_tilt.txt: per tilt, the largest dispatch of every search3_box instantiation, all counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05}
TILTS=${2:-"0 3 45"}
OUT=gpurun_out/${TAG}_pmc_tilt.txt
: > $OUT
PA="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
PB="SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES"
PC="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_DCACHE_REQ SQC_DCACHE_MISSES"
for T in $TILTS; do
  for P in A B C; do
    eval "CS=\$P$P"
    D=gpurun_out/${TAG}_pt_${T}_$P
    rocprofv3 --pmc $CS -d $D -o q --output-format csv -- python3 scripts/prof_counters.py 4096 --tilt=$T > $D.log 2>&1
    echo "== tilt $T pass $P ($CS)" >> $OUT
    python3 scripts/pmc_summary.py $D search3_box >> $OUT 2>&1
    rm -rf $D
  done
done
tail -5 $OUT
