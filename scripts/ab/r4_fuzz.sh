cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/fuzz_bitparity.py 9000:9600 > gpurun_out/r4_fuzz.log 2>&1; echo rc=$?; tail -3 gpurun_out/r4_fuzz.log | cut -c1-300
python scripts/fuzz_bitparity.py 9000:9400 - flatearth > gpurun_out/r4_fuzz_fe.log 2>&1; echo rc=$?; tail -3 gpurun_out/r4_fuzz_fe.log | cut -c1-300
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "blocked" 2>&1 | tail -2
