# partial measurement set for a revision that only touches the 40-dof (cassie_tray_box.xml) instantiation: GPU suite,
# config 5 bench line, its rocprofv3 kernel stats; plus the headline bench line as the unchanged-kernel control
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.log 2>&1
timeout 300 python bench.py --model cassie_tray_box > gpurun_out/bench_tray.json 2> gpurun_out/bench_tray.err
timeout 600 python bench.py > gpurun_out/bench_cassie.json 2> gpurun_out/bench_cassie.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tray -- python $R/bench.py --model cassie_tray_box --steps 200 --warmup 50 > $R/gpurun_out/prof_tray.log 2>&1
cd $R
tail -4 gpurun_out/pytest_gpu.log
for f in tray cassie; do echo "== $f"; tail -c 2500 gpurun_out/bench_$f.json; tail -3 gpurun_out/bench_$f.err; done
find gpurun_out/prof_tray -name "*kernel_stats.csv" | head -1 | xargs head -4
