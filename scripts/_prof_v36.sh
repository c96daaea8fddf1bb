cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for V in 3 6; do
D=gpurun_out/r05_v${V}_pmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $D -o q --output-format csv -- python3 scripts/prof_counters.py 4096 --version=$V > $D.log 2>&1
echo "== version $V" >> gpurun_out/r05_v36_pmc.txt
python3 scripts/pmc_summary.py $D search3_box >> gpurun_out/r05_v36_pmc.txt 2>&1
rm -rf $D
done
cat gpurun_out/r05_v36_pmc.txt
