# headline profile set: kernel stats, SQ PMC pass, FETCH / WRITE passes (separate passes, as the guide prescribes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02}
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -o s --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/${TAG}_pmcS -o q --output-format csv -- $B > gpurun_out/${TAG}_pmcS.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_pmcF -o f --output-format csv -- $B > gpurun_out/${TAG}_pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmcW -o w --output-format csv -- $B > gpurun_out/${TAG}_pmcW.log 2>&1
ls gpurun_out/${TAG}_stats gpurun_out/${TAG}_pmcS
tail -2 gpurun_out/${TAG}_pmcS.log | cut -c1-300
