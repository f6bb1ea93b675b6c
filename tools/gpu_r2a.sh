# round 2, first GPU call: the whole -m gpu suite, the bench line of every config, a short-run bench (representativeness),
# rocprofv3 kernel stats of the 32-dof and the 40-dof instantiations
mkdir -p gpurun_out; nproc > gpurun_out/nproc.txt
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.log 2>&1
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_cassie.json 2> gpurun_out/bench_cassie.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd --no-other-mode > gpurun_out/bench_cassie_short.json 2> gpurun_out/bench_cassie_short.err
timeout 300 python bench.py --model cassie_hfield > gpurun_out/bench_hfield.json 2> gpurun_out/bench_hfield.err
timeout 300 python bench.py --model cassie_tray_box > gpurun_out/bench_tray.json 2> gpurun_out/bench_tray.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cassie -- python $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-step-pd --no-other-mode > $R/gpurun_out/prof_cassie.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tray -- python $R/bench.py --model cassie_tray_box --steps 200 --warmup 50 > $R/gpurun_out/prof_tray.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hfield -- python $R/bench.py --model cassie_hfield --steps 200 --warmup 50 > $R/gpurun_out/prof_hfield.log 2>&1
cd $R
tail -8 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cat gpurun_out/single_sim_rate.txt
for f in cassie cassie_short hfield tray; do echo "== $f"; tail -c 3000 gpurun_out/bench_$f.json; tail -3 gpurun_out/bench_$f.err; done
