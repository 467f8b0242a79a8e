cd $GRAFT_REPO_ROOT
for round in 1 2; do
python scripts/ab/blk_check.py 100000 scripts/ab/blk0.so plain 2>&1 | grep -v amdgpu.ids
python scripts/ab/blk_check.py 100000 pygenray_amd/csrc/libpgr_hip.so plain,blocked,end_state 2>&1 | grep -v amdgpu.ids
done
python scripts/regress.py --check scripts/regress_ref.json 2>&1 | tail -1
