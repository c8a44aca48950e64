cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03_r; mkdir -p $O; rm -f $O/pmc_sq.txt
i=0
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
i=$((i+1))
rm -rf /tmp/pmc_$i
MULLS_SPLIT_MAX_PAIRS=0 timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$i -- python tools/gpu_one.py 3 4096 2 > $O/pmc_run_$i.log 2>&1
echo "== $c" >> $O/pmc_sq.txt
python tools/pmc_summary.py /tmp/pmc_$i | grep -E "k_cert|k_nn_lds|k_accum<102|k_tgt" >> $O/pmc_sq.txt 2>&1
done
cat $O/pmc_sq.txt; tail -3 $O/pmc_run_3.log
