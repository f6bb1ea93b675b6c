mkdir -p gpurun_out/pmc2; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 100 --warmup 50 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc2/p1 -- $CMD > $R/gpurun_out/pmc2/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/pmc2/p2 -- $CMD > $R/gpurun_out/pmc2/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc2/p3 -- $CMD > $R/gpurun_out/pmc2/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc2/p4 -- $CMD > $R/gpurun_out/pmc2/p4.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for p in sorted(glob.glob('gpurun_out/pmc2/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'cassie_step' in r.get('Kernel_Name',''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(p.split('/')[2], {k:(sum(v)/len(v)) for k,v in acc.items()})
PY
tail -2 gpurun_out/pmc2/p2.log
