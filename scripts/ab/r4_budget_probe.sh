cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/bp_a -- python3 $R/scripts/isa_budget_run.py headline quiet 0 > $R/gpurun_out/bp_a.log 2>&1; echo rc=$?
tail -3 $R/gpurun_out/bp_a.log | cut -c1-400
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/bp_b -- python3 $R/scripts/isa_budget_run.py headline quiet 0 > $R/gpurun_out/bp_b.log 2>&1; echo rc=$?
tail -3 $R/gpurun_out/bp_b.log | cut -c1-400
ls $R/gpurun_out/bp_a/*/ $R/gpurun_out/bp_b/*/
