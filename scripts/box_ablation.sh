# Level-0 ablations of the box kernel (CVHIP_DEBUG, cvhip_internal.hpp): time of the search class per step with parts of the
# kernel switched off (results are then wrong on purpose).  usage: bash scripts/box_ablation.sh > gpurun_out/ablation.txt
cd ${GRAFT_REPO_ROOT:-.}
for d in 0 256 512 8 64 16 72 88; do
  CVHIP_DEBUG=$d python3 bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('debug', $d, 'step', d['ms_per_step'], 'search', d['kernel_ms_per_step']['search'])"
done
