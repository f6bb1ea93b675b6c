mkdir -p gpurun_out/pmcx; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for MODE in exact-pd drive-pd; do
CMD="python $R/bench.py --mode $MODE --steps 100 --warmup 50 --no-cpu-baseline --no-step-pd --no-other-mode --parity-envs 4"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcx/${MODE}_$c -- $CMD > /dev/null 2>&1
done
done
cd $R
python - <<'PY'
import csv, glob, collections
for p in sorted(glob.glob('gpurun_out/pmcx/*/*/*counter_collection.csv')):
    v=[float(r['Counter_Value']) for r in csv.DictReader(open(p)) if 'cassie_step' in r.get('Kernel_Name','')]
    v=sorted(v)[len(v)//3:]
    print(p.split('/')[2], sum(v)/len(v), 'KB', len(v))
PY
