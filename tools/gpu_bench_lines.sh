O=gpurun_out/r03_final; mkdir -p $O
timeout 400 python bench.py --total-pairs 1024 --no-cpu-baseline > $O/bench_tp1024.json 2> $O/bench_tp1024.err
timeout 400 python bench.py --total-pairs 128 --no-cpu-baseline > $O/bench_tp128.json 2> $O/bench_tp128.err
timeout 400 python bench.py --data demo --no-cpu-baseline > $O/bench_demo.json 2> $O/bench_demo.err
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
for f in ('bench','bench_tp1024','bench_tp128','bench_demo'):
    d=json.loads(open('gpurun_out/r03_final/%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['ms_per_step'],2), d['untimed_priming_steps'], round(d['roofline']['frac'],4), round(d.get('value_converging',{}).get('value',0)), round(d.get('value_sustained',{}).get('value',0)), round(d.get('value_end_to_end',{}).get('value',0)))
PY
