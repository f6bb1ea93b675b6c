# round 3, first GPU call: the whole -m gpu suite (new: drive-pd parity net, config-3 shapes, stress variant, goldens), the
# bench line of every config with the 10-region timing, the driver's short command, the 65536-env total on one GPU,
# rocprofv3 kernel stats of the headline kernel
mkdir -p gpurun_out; nproc > gpurun_out/nproc.txt
(time timeout 2400 python -m pytest tests -m gpu -x -q -s) > gpurun_out/pytest_gpu.log 2>&1
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
(time timeout 600 python bench.py) > gpurun_out/bench_cassie.json 2> gpurun_out/bench_cassie.err
(time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/bench_cassie_short.json 2> gpurun_out/bench_cassie_short.err
timeout 300 python bench.py --model cassie_hfield > gpurun_out/bench_hfield.json 2> gpurun_out/bench_hfield.err
timeout 300 python bench.py --model cassie_tray_box > gpurun_out/bench_tray.json 2> gpurun_out/bench_tray.err
timeout 300 python bench.py --total-envs 65536 --steps 100 --warmup 50 --repeats 5 > gpurun_out/bench_total65536.json 2> gpurun_out/bench_total65536.err
timeout 300 python bench.py --envs-per-gpu 8192 --steps 200 --warmup 50 --force-collectives --no-cpu-baseline --no-step-pd --no-other-mode > gpurun_out/bench_8192_collectives.json 2> gpurun_out/bench_8192_collectives.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cassie -- python $R/bench.py --steps 200 --warmup 50 --repeats 3 --no-cpu-baseline --no-step-pd --no-other-mode > $R/gpurun_out/prof_cassie.log 2>&1
cd $R
tail -15 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cat gpurun_out/single_sim_rate.txt
for f in cassie cassie_short hfield tray total65536 8192_collectives; do echo "== $f"; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$f.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "value_min", "value_max", "max_qpos_err", "ms_per_step", "value_exact_pd", "value_all_outputs_every_substep", "value_step_pd", "scaling")})
    print(d["config"]["workload"][:120], d["parity"].get("ranks_compared"), d["parity"]["frac_envs_with_equal_ncon_nefc_iters"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
except Exception as e:
    print("no line:", e)
PY
tail -3 gpurun_out/bench_$f.err; done
