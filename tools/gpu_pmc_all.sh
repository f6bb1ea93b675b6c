# PMC passes of the bench's dominant kernel (separate --pmc runs with --kernel-trace only, as gpurun requires), then
# tools/pmc_summary.py reduces the CSVs to gpurun_out/pmc_summary.json.  MODE=exact-pd|drive-pd (default drive-pd).
mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py ${MODEL:+--model $MODEL} --mode ${MODE:-drive-pd} --streams 1 --steps 100 --warmup 50 --no-cpu-baseline --no-step-pd --no-other-mode --parity-envs 4"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM" "SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_F64" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $i " && continue   # (ONLY="1 8 9": the totals and the instruction-class passes)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc/p$i -- $CMD > $R/gpurun_out/pmc/p$i.log 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc_summary.json
cat gpurun_out/pmc_summary.json | head -70
