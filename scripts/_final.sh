cd $GRAFT_REPO_ROOT
python scripts/ransac_live_fraction.py 2>/dev/null | tail -1
bash scripts/_run_prof.sh r02d > gpurun_out/r02d_prof.log 2>&1
tail -c 300 gpurun_out/r02d_bench.json
