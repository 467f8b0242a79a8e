cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -s > gpurun_out/r4_gpu_suite.log 2>&1
rc=$?
grep -v "^$" gpurun_out/r4_gpu_suite.log | tail -15 | cut -c1-300
exit $rc
