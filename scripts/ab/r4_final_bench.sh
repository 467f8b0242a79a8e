cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; echo main rc=$?
python bench.py --flat-earth --no-cpu-baseline --no-eigenray --no-legs > gpurun_out/r04_bench_line_flatearth.json 2>/dev/null; echo fe rc=$?
python bench.py --range-dependent --no-cpu-baseline --no-eigenray --no-legs > gpurun_out/r04_bench_line_config2.json 2>/dev/null; echo rd rc=$?
python bench.py --range-dependent --blocked --no-cpu-baseline --no-eigenray --no-legs > gpurun_out/r04_bench_line_config2_blocked.json 2>/dev/null; echo rdb rc=$?
python -c "
import json
d=json.load(open('gpurun_out/r04_bench_line.json'))
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'],'traffic',d['roofline']['traffic_gb_per_launch'],'valu',d.get('roofline_valu',{}).get('frac'))
for k,v in d['legs'].items():
    print(k, {m:(round(x['kernel_ms'],3), round(x['frac'],3)) for m,x in v.items() if isinstance(x,dict) and 'kernel_ms' in x} if k!='api' else {m:{a:round(b,2) if isinstance(b,float) else b for a,b in x.items()} for m,x in v.items() if isinstance(x,dict)})
print('lone',d['lone_wave_ms']); print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['cores'],d['cpu_baseline_c']['value']); print('eig',{k:v for k,v in d['eigenray'].items() if k!='config'})
"
