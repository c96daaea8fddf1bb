cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_orb_ransac_gpu.py tests/test_sfm3_gpu.py tests/test_host_cpp_gpu.py -m gpu -x -q 2>&1 | tail -3 &&
python bench.py --config sfm3 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r02f_sfm_bench.json &&
rocprofv3 --kernel-trace --stats -d gpurun_out/r02f_sfm_stats -o s --output-format csv -- python3 bench.py --config sfm3 --steps 2 --warmup 1 > gpurun_out/r02f_sfm_stats.log 2>&1
cat gpurun_out/r02f_sfm_bench.json
