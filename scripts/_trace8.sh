cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for s in 0/2 1/4 3/8; do python bench.py --simulate-shard $s --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$s', d['ms_per_step'], d['kernel_ms_per_step'])"; done
rocprofv3 --kernel-trace -d gpurun_out/r2_trace8 -o t --output-format csv -- python3 bench.py --simulate-shard 3/8 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r2_trace8.log 2>&1
