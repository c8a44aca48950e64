#!/usr/bin/env bash
# The whole GPU check of a build on the MI355X box: tools/gpu_full_check.sh <tag>  ->  gpurun_out/<tag>/{pytest.txt, modes.txt, bench.json}
set -u
O=gpurun_out/${1:-check}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python tools/gpu_modes.py 1 8 32 64 128 192 256 384 512 1024 2048 4096 > $O/modes.txt 2>&1; cat $O/modes.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python - "$O" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print("value %.0f  ms/step %.2f  roofline frac %.4f  converging %.0f  sustained %.0f  end to end %.0f" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["value_converging"]["value"], d["value_sustained"]["value"], d["value_end_to_end"]["value"]))
PY
