cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PGR_BENCH_ONE_GPU=1 timeout -k 10 500 python bench.py --gpus 4 --backend gloo --steps 3 --warmup 1 --eigen-rays 1000000 > gpurun_out/r4_rehearsal_4ranks.json 2> gpurun_out/r4_rehearsal_4ranks.err
echo rc=$?
cut -c1-3000 gpurun_out/r4_rehearsal_4ranks.json
tail -5 gpurun_out/r4_rehearsal_4ranks.err
