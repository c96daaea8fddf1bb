# Where the box kernels' VALU instructions go: SQ counters of the level-0 launches with parts of the kernel switched off
# (-DCVHIP_ABLATIONS build, made on the GPU box; CVHIP_DEBUG bits: 256 setup only, 512 no staging, 8 no walk, 64 no hit
# branch, 16 no exact phase).   gpurun -- 'bash scripts/_prof_ablate_pmc.sh r05 "0 3"'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05}
TILTS=${2:-"0 3"}
OUT=gpurun_out/${TAG}_ablate_pmc.txt
CVHIP_EXTRA_FLAGS=-DCVHIP_ABLATIONS python3 -m cybervision_amd.build --force > gpurun_out/${TAG}_ablate_build.log 2>&1 || exit 1
: > $OUT
for T in $TILTS; do
  for D in 0 256 8 64 16; do
    export CVHIP_DEBUG=$D
    DIR=gpurun_out/${TAG}_ab_${T}_$D
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $DIR -o q --output-format csv -- python3 scripts/prof_counters.py 4096 --tilt=$T > $DIR.log 2>&1
    echo "== tilt $T debug $D" >> $OUT
    python3 scripts/pmc_summary.py $DIR search3_box >> $OUT 2>&1
    rm -rf $DIR
  done
done
unset CVHIP_DEBUG
tail -3 $OUT
