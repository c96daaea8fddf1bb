cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in w6 w7; do cp tmp_ab/$v.so cybervision_amd/libcvhip.so; python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['ms_per_step'], d['kernel_ms_per_step']['search'])"; done; done
