mkdir -p gpurun_out
# config 5 (cassie_tray_box.xml): bench line + the GPU parity tests that touch the 40-dof kernel
run() { python bench.py --steps ${STEPS:-1000} --warmup 100 --no-cpu-baseline --no-step-pd --no-other-mode --parity-envs 16 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2: %.3f M  kernel %.3f ms  err %.1e rows %.1f iters %.1f' % (d['value']/1e6, d['roofline']['kernel_ms'], d['max_qpos_err'], d['mean_constraint_rows'], d['mean_pgs_iterations']))"; }
run tray "--model cassie_tray_box"
run tray8k "--model cassie_tray_box --envs-per-gpu 8192"
run cassie ""
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
