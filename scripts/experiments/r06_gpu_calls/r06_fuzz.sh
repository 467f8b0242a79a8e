#!/bin/bash
# round 6: random sweeps with NEW seeds (one gpurun call, product binary) -> profiles/r06_fuzz_sweeps.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06_fuzz; mkdir -p $O
{
echo "# random sweeps of round 6 (one gpurun call, seeds no earlier sweep used): scripts/fuzz_bitparity.py 40000:42000 ; 40000:41000 - flatearth ; scripts/fuzz_blocked.py 41000:42200 ; scripts/fuzz_persistent.py 43000:43120 ; 43000:43080 flatearth"
timeout -k 10 900 python scripts/fuzz_bitparity.py 40000:42000 2>&1 | tail -n 2
timeout -k 10 600 python scripts/fuzz_bitparity.py 40000:41000 - flatearth 2>&1 | tail -n 2
timeout -k 10 600 python scripts/fuzz_blocked.py 41000:42200 2>&1 | tail -n 2
timeout -k 10 600 python scripts/fuzz_persistent.py 43000:43120 2>&1 | tail -n 1
timeout -k 10 600 python scripts/fuzz_persistent.py 43000:43080 flatearth 2>&1 | tail -n 1
python -c "
import sys; sys.path.insert(0, '.')
from pygenray_amd import _lib
print('# device_code_sha256', _lib.device_code_sha256())"
} > $O/fuzz_sweeps.txt 2>&1
cat $O/fuzz_sweeps.txt
# (the 1e6-ray trajectory line once more, now that profiles/r06_traffic.json holds its PMC pass: the line then carries `traffic`)
timeout -k 10 300 python bench.py --rays 1000000 --no-cpu-baseline --no-eigenray --no-legs > $O/bench_line_rays_1e6_traj.json 2> $O/bench.err; echo "bench 1e6 traj rc $?"
