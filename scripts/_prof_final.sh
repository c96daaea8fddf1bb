# final measurement set of a round: headline profiles (scripts/_run_prof.sh), the full bench line, config-5 kernel stats,
# emulated band shares.   gpurun -- 'CVHIP_GIT_HEAD=<hash> bash scripts/_prof_final.sh r03'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03}
bash scripts/_run_prof.sh $TAG
python3 bench.py --steps 25 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --config sfm3 --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_sfm3.json 2>> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_sfm_stats -o s --output-format csv -- python3 bench.py --config sfm3 --steps 3 --warmup 1 > gpurun_out/${TAG}_sfm_stats.log 2>&1
find gpurun_out/${TAG}_sfm_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_sfm3_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_sfm_stats
rm -f gpurun_out/${TAG}_simulated_shards.jsonl gpurun_out/${TAG}_rehearsal_n2_n5.txt
for s in 0/2 1/4 3/8; do python3 bench.py --simulate-shard $s --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/${TAG}_simulated_shards.jsonl; done
tail -c 400 gpurun_out/${TAG}_bench.json
# round 5: the stepped kernels' SQ / SQC counters on tilted pairs, the gloo rehearsal of the N-rank path on one GPU
bash scripts/_prof_tilt.sh $TAG "0 3 10 45" > /dev/null 2>&1
for N in 2 5; do
  CVHIP_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "rehearsal|sharded_equals" | cut -c1-400 >> gpurun_out/${TAG}_rehearsal_n2_n5.txt
done
tail -3 gpurun_out/${TAG}_rehearsal_n2_n5.txt | cut -c1-200
