mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQC?_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|GRBM_[A-Z_]+|TCC_[A-Z_0-9]+)\b" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters.txt
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 100 --warmup 50 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc/p1 -- $CMD > $R/gpurun_out/pmc/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc/p2 -- $CMD > $R/gpurun_out/pmc/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/p3 -- $CMD > $R/gpurun_out/pmc/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/p4 -- $CMD > $R/gpurun_out/pmc/p4.log 2>&1
cd $R; ls gpurun_out/pmc/*; grep -c ICACHE gpurun_out/pmc/counters.txt
python - <<'PY'
import csv, glob, collections
for p in sorted(glob.glob('gpurun_out/pmc/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'cassie_step' in r.get('Kernel_Name',''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(p.split('/')[2], {k:(sum(v)/len(v)) for k,v in acc.items()}, 'n', {k:len(v) for k,v in acc.items()})
PY
