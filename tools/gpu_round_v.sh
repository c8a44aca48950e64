O=gpurun_out/r03_v; mkdir -p $O; rm -f $O/sweep.txt
for h in 0.6 0.8 1.0 1.3 1.7; do echo "== GRID_H0 $h" >> $O/sweep.txt; MULLS_GRID_H0=$h timeout 300 python tools/gpu_modes.py 4096 >> $O/sweep.txt 2>&1; done
for c in 16 32 128; do echo "== CERT_SMALL n/a; slack_min sweep skipped" > /dev/null; done
for r in "0.02 0.5 1.0" "0.05 1.0 2.0" "0.01 0.3 0.5"; do set -- $r; echo "== slack min $1 max $2 rate $3" >> $O/sweep.txt; MULLS_CERT_SLACK_MIN=$1 MULLS_CERT_SLACK_MAX=$2 MULLS_CERT_SLACK_RATE=$3 timeout 300 python tools/gpu_modes.py 4096 >> $O/sweep.txt 2>&1; done
cat $O/sweep.txt
