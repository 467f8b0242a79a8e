cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( while true; do sleep 60; echo "heartbeat $(date +%T)"; done ) &
HB=$!
(echo "# scripts/bitparity.py - 1 1 - 1000000 11: ALL 1 000 000 rays of the configs[3] / configs[4] fan (11 samples each) against oracle.MATH_CR, round-4 binary (Ziv powers)"; python scripts/bitparity.py - 1 1 - 1000000 11 2>&1 | grep -v amdgpu.ids; python -c "from pygenray_amd import _lib; print('# device_code_sha256', _lib.device_code_sha256())") > gpurun_out/r04_bitparity_1e6_rays.txt
kill $HB
tail -6 gpurun_out/r04_bitparity_1e6_rays.txt | cut -c1-300
