cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2; do for S in "0.02,0.10,1.0" "0.02,0.20,2.0" "0.05,0.20,1.0" "0.01,0.05,0.5" "0.05,0.30,2.0" "0.03,0.15,1.5"; do
MULLS_CERT_SLACK=$S python bench.py --no-other-configs --no-cpu-baseline --no-end-to-end --no-converging --steps 10 --sustain-s 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernel_ms_per_step']
print('slack $S  value %.1f k | search %.2f' % (j['value']/1e3, k['ms_nn']))"
MULLS_CERT_SLACK=$S python bench.py --config 0 --no-other-configs --no-cpu-baseline --no-end-to-end --no-converging --steps 5 --sustain-s 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernel_ms_per_step']
print('   cfg0 4096 $S  value %.1f k | search %.2f' % (j['value']/1e3, k['ms_nn']))"
done; done
