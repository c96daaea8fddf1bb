set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t5.log 2>&1; tail -4 gpurun_out/r2_t5.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b1.log 2>&1; tail -c 1500 gpurun_out/r2_b1.log
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r2_pmcF1 -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_pmcF1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r2_pmcW1 -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_pmcW1.log 2>&1
ls gpurun_out/r2_pmcF1 | head
