O=gpurun_out/r03_j; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
b=json.loads(open('gpurun_out/r03_j/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], json.dumps(b['value_end_to_end']), json.dumps(b['roofline']['kernel_ms_per_step']), b.get('value_converging',{}).get('value'))
PY
