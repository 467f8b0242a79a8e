cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
for lib in scripts/ab/noziv.so pygenray_amd/csrc/libpgr_hip.so; do
python scripts/kbench.py --lib $lib --reps 8 --modes nosave sample 2>&1 | grep -v amdgpu.ids | sed "s#^#$(basename $lib) #" | cut -c1-150
done
python scripts/kbench.py --lib scripts/ab/noziv.so --reps 5 --modes nosave --amin -20 --amax -19.99936 --rays 64 2>&1 | grep -v amdgpu | sed "s#^#noziv lone #" | cut -c1-150
python scripts/kbench.py --reps 5 --modes nosave --amin -20 --amax -19.99936 --rays 64 2>&1 | grep -v amdgpu | sed "s#^#ziv lone #" | cut -c1-150
done
python scripts/regress.py --check scripts/regress_ref.json 2>&1 | tail -1
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "hard_cases or building_blocks" 2>&1 | tail -2
