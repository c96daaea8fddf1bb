# as box_ablation.sh, for the stepped instantiations: level-0 time of a tilted pair (default 10 degrees) with parts of the
# box kernel switched off (-DCVHIP_ABLATIONS build).  usage: bash scripts/box_ablation_tilt.sh [tilt] > gpurun_out/ablation_tilt.txt
cd ${GRAFT_REPO_ROOT:-.}
T=${1:-10}
for d in 0 256 512 8 64 16; do
  echo -n "debug $d: "; CVHIP_DEBUG=$d python3 scripts/sweep_quick.py $T 2>/dev/null | tail -1
done
