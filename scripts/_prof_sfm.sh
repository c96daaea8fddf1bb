cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_sfm_stats -o s --output-format csv -- python3 bench.py --config sfm3 --steps 2 --warmup 1 > gpurun_out/r02_sfm_stats.log 2>&1
tail -1 gpurun_out/r02_sfm_stats.log | cut -c1-900
