# effective shader clock under the step kernel: GRBM_GUI_ACTIVE (cycles the GPU was busy) / the kernel's duration
mkdir -p gpurun_out/pmc_clock; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_clock -- python $R/bench.py --mode ${MODE:-drive-pd} --steps 100 --warmup 50 --no-cpu-baseline --no-step-pd --no-other-mode --parity-envs 4 > $R/gpurun_out/pmc_clock/run.log 2>&1
cd $R
python - <<'PY'
import csv, glob
cnt = glob.glob('gpurun_out/pmc_clock/**/*counter_collection.csv', recursive=True)
tr = glob.glob('gpurun_out/pmc_clock/**/*kernel_trace.csv', recursive=True)
dur = {}
for r in csv.DictReader(open(tr[0])):
    if 'cassie_step_kernel' in r['Kernel_Name']:
        dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
act = {}
for r in csv.DictReader(open(cnt[0])):
    if 'cassie_step_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        act[r['Dispatch_Id']] = act.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
ids = [i for i in dur if i in act][-20:]
for i in ids[-5:]:
    print("dispatch %s: %.3f ms, GRBM_GUI_ACTIVE %.0f -> %.3f GHz" % (i, dur[i] / 1e6, act[i], act[i] / dur[i]))
import statistics
print("mean effective clock over %d launches: %.3f GHz" % (len(ids), statistics.mean(act[i] / dur[i] for i in ids)))
PY
