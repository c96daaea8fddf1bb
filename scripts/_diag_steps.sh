# displacement steps per wave of the box kernels at level 0 (ablation build, CVHIP_DEBUG=32: counter 1 = sum over waves of
# steps x staged planes, counter 2 = waves that walked, counter 3 = declined workgroups)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CVHIP_EXTRA_FLAGS=-DCVHIP_ABLATIONS python3 -m cybervision_amd.build --force > /dev/null 2>&1 || exit 1
export CVHIP_DEBUG=32
for T in 0 3 10 30; do echo "tilt $T"; python3 scripts/prof_counters.py 4096 --tilt=$T --count 2>/dev/null | tail -1; done
