cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/r2_trace512 -o t --output-format csv -- python3 bench.py --size 512 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r2_trace512.log 2>&1
ls gpurun_out/r2_trace512
