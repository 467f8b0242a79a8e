#!/bin/bash
# Evidence, part B (one gpurun call, ~16 min): all-rays bit parity against the oracle with the binary as built.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-ev}
cd $R
O=gpurun_out/${TAG}_b; mkdir -p $O
SHA=$(python -c "import sys; sys.path.insert(0,'.'); from pygenray_amd import _lib; print(_lib.device_code_sha256())")
echo "# scripts/bitparity.py - 1 <config> - 100000 1001 default-form on the GPU box: ALL 100 000 rays x 1001 samples of bench.py's three trajectory workloads (its own tables) against oracle.MATH_CR" > $O/bitparity_S1001.txt
for C in ${CONFIGS:-11 12 13}; do
  python scripts/bitparity.py - 1 $C - 100000 1001 default-form >> $O/bitparity_S1001.txt 2>&1; echo "config $C rc $?"
  echo >> $O/bitparity_S1001.txt
done
echo "# device_code_sha256 $SHA" >> $O/bitparity_S1001.txt
echo "# scripts/bitparity.py - 1 1 - 1000000 11 on the GPU box: ALL 1 000 000 rays of the configs[3] / configs[4] fan (persistent waves) against oracle.MATH_CR" > $O/bitparity_1e6_rays.txt
# (the oracle needs ~9 minutes for 1e6 rays and prints nothing meanwhile: a heartbeat file keeps the box's silence watchdog quiet)
( while sleep 60; do date >> $O/heartbeat.txt; done ) &
HB=$!
python scripts/bitparity.py - 1 1 - 1000000 11 >> $O/bitparity_1e6_rays.txt 2>&1; echo "1e6 rc $?"
kill $HB
echo "# device_code_sha256 $SHA" >> $O/bitparity_1e6_rays.txt
grep -c "1.00000" $O/bitparity_S1001.txt; tail -n 4 $O/bitparity_1e6_rays.txt | cut -c1-250
