#!/bin/bash
# round 6: ALL rays of the three benched trajectory workloads (1e5 x 1001 samples each) and ALL 1e6 rays of the configs[3] / [4] fan against
# the oracle (MATH_CR), with this round's library -> profiles/r06_bitparity_S1001.txt, profiles/r06_bitparity_1e6_rays.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06_bp; mkdir -p $O
( while true; do date >> $O/heartbeat.txt; sleep 60; done ) &
HB=$!
SHA=$(python -c "
import sys; sys.path.insert(0, '.')
from pygenray_amd import _lib
print(_lib.device_code_sha256())")
echo "# scripts/bitparity.py - 1 <config> - 100000 1001 default-form on the GPU box: ALL 100 000 rays x 1001 samples of bench.py's three trajectory workloads (its own tables) against oracle.MATH_CR" > $O/bitparity_S1001.txt
for C in 11 12 13; do
  timeout -k 10 900 python scripts/bitparity.py - 1 $C - 100000 1001 default-form >> $O/bitparity_S1001.txt 2>&1; echo "config $C rc $?"
  echo >> $O/bitparity_S1001.txt
done
echo "# device_code_sha256 $SHA" >> $O/bitparity_S1001.txt
echo "# scripts/bitparity.py - 1 1 - 1000000 11 on the GPU box: ALL 1 000 000 rays of the configs[3] / configs[4] fan (persistent waves) against oracle.MATH_CR" > $O/bitparity_1e6_rays.txt
timeout -k 10 1000 python scripts/bitparity.py - 1 1 - 1000000 11 >> $O/bitparity_1e6_rays.txt 2>&1; echo "1e6 rc $?"
echo "# device_code_sha256 $SHA" >> $O/bitparity_1e6_rays.txt
kill $HB
grep -c "1.00000" $O/bitparity_S1001.txt; tail -n 8 $O/bitparity_1e6_rays.txt | cut -c1-250
