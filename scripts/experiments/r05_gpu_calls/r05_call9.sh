cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05c9; mkdir -p $O
python scripts/fuzz_blocked.py > $O/fuzz_blocked.log 2>&1; echo "fuzz_blocked rc $?"; tail -n 2 $O/fuzz_blocked.log
python scripts/fuzz_bitparity.py 11000:11500 > $O/fuzz_a.log 2>&1; echo "fuzz rc $?"; tail -n 3 $O/fuzz_a.log
python scripts/fuzz_bitparity.py 11000:11300 - flatearth > $O/fuzz_b.log 2>&1; echo "fuzz fe rc $?"; tail -n 3 $O/fuzz_b.log
