O=gpurun_out/r03_k; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
b=json.loads(open('gpurun_out/r03_k/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], json.dumps(b['value_end_to_end'])[:400], json.dumps(b['roofline']['kernel_ms_per_step']), b.get('value_converging',{}).get('value'), b.get('value_sustained'))
PY
