O=gpurun_out/r03_o; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
for f in 0 1; do
echo "== FUSED_TGT_SETUP $f" >> $O/modes.txt
MULLS_FUSED_TGT_SETUP=$f timeout 600 python tools/gpu_modes.py 1 128 1024 4096 >> $O/modes.txt 2>&1
done
cat $O/modes.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
