cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh prof_r04 > gpurun_out/prof_r04.log 2>&1; echo profiles rc=$?
bash scripts/collect_isa_budget.sh budget_r04 > gpurun_out/budget_r04.log 2>&1; echo budget rc=$?
tail -2 gpurun_out/prof_r04.log gpurun_out/budget_r04.log
