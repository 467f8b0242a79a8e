set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r04_bitparity_S1001.txt
echo "# scripts/bitparity.py - 1 <config> - 100000 1001 default-form on the GPU box: ALL 100 000 rays x 1001 samples of bench.py's three trajectory workloads (its own tables) against oracle.MATH_CR" > $out
for c in 11 12 13; do
  python scripts/bitparity.py - 1 $c - 100000 1001 default-form >> $out 2>&1
  echo >> $out
  echo "config $c done"
done
python -c "from pygenray_amd import _lib; print('# device_code_sha256', _lib.device_code_sha256())" >> $out
