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
# sparse stage + config 5 (bench.py --config sfm3): kernel stats and the two traffic passes
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_sfm_stats -o s --output-format csv -- python3 bench.py --config sfm3 --steps 2 --warmup 1 > gpurun_out/${TAG}_sfm_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_sfm_pmcF -o f --output-format csv -- python3 bench.py --config sfm3 --steps 1 --warmup 0 > gpurun_out/${TAG}_sfm_pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_sfm_pmcW -o w --output-format csv -- python3 bench.py --config sfm3 --steps 1 --warmup 0 > gpurun_out/${TAG}_sfm_pmcW.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/${TAG}_sfm_pmcS -o q --output-format csv -- python3 bench.py --config sfm3 --steps 1 --warmup 0 > gpurun_out/${TAG}_sfm_pmcS.log 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --config sfm3 --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_sfm3.json 2>> gpurun_out/${TAG}_bench.err
for s in 0/2 1/4 3/8; do python3 bench.py --simulate-shard $s --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/${TAG}_simulated_shards.jsonl; done
tail -c 600 gpurun_out/${TAG}_bench.json
