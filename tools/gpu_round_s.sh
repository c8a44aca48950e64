O=gpurun_out/r03_s; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python tools/gpu_modes.py 1 8 32 64 128 192 256 384 512 1024 2048 4096 > $O/modes.txt 2>&1; cat $O/modes.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_s/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d['value_converging']['value'], d['value_sustained']['value'], d['value_end_to_end']['value'])
PY
