set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/regress.py --check scripts/regress_ref.json > gpurun_out/r4_regress_clean.log 2>&1
tail -2 gpurun_out/r4_regress_clean.log
python -m pytest tests/test_hip_parity.py -x -q -m gpu -s -k "as_benchmarked or wide_pins or golden" > gpurun_out/r4_new_tests.log 2>&1 || { tail -40 gpurun_out/r4_new_tests.log; exit 1; }
tail -5 gpurun_out/r4_new_tests.log
python bench.py > gpurun_out/r4_bench_clean.json 2> gpurun_out/r4_bench_clean.err
cat gpurun_out/r4_bench_clean.json | cut -c1-1500
