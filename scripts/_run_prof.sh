# headline profile set: kernel stats, SQ PMC pass, FETCH / WRITE passes (separate passes, as the guide prescribes),
# then the summaries the bench line reads (profiles/current_*.json are copies of <TAG>_pmc.json / <TAG>_traffic.json)
#   gpurun -- 'CVHIP_GIT_HEAD=<hash> bash scripts/_run_prof.sh r03'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03}
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-count-step"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -o s --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/${TAG}_pmcS -o q --output-format csv -- $B > gpurun_out/${TAG}_pmcS.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_pmcF -o f --output-format csv -- $B > gpurun_out/${TAG}_pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmcW -o w --output-format csv -- $B > gpurun_out/${TAG}_pmcW.log 2>&1
# three identical bench steps per profiled run (--no-count-step): one warm-up step, the timed step, the instrumented step
python3 scripts/collect_pmc.py gpurun_out/${TAG}_pmcS gpurun_out/${TAG}_pmc.json 3 > gpurun_out/${TAG}_collect.log 2>&1
python3 scripts/collect_traffic.py gpurun_out/${TAG}_pmcF gpurun_out/${TAG}_pmcW gpurun_out/${TAG}_traffic.json 3 >> gpurun_out/${TAG}_collect.log 2>&1
cp gpurun_out/${TAG}_stats/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null || find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
# the bulky raw counter dumps do not need to travel back
rm -rf gpurun_out/${TAG}_pmcS gpurun_out/${TAG}_pmcF gpurun_out/${TAG}_pmcW gpurun_out/${TAG}_stats
tail -4 gpurun_out/${TAG}_collect.log | cut -c1-300
