cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1; echo smoke rc=$?; tail -2 gpurun_out/r4_smoke.log
python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_suite_final.log 2>&1
rc=$?
grep -v "^$" gpurun_out/r4_gpu_suite_final.log | tail -6 | cut -c1-300
exit $rc
