mkdir -p gpurun_out
run() { python bench.py --mode $1 --steps ${STEPS:-1000} --warmup 100 --no-cpu-baseline --no-step-pd --no-other-mode --parity-envs 16 $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2 $1 $3: %.3f M  kernel %.3f ms  err %.1e rows %.1f iters %.1f' % (d['value']/1e6, d['roofline']['kernel_ms'], d['max_qpos_err'], d['mean_constraint_rows'], d['mean_pgs_iterations']))"; }
for mode in exact-pd drive-pd; do
  CASSIE_NO_BALANCE=1 run $mode nobalance
  run $mode balance
done
run drive-pd balance "--envs-per-gpu 8192"
CASSIE_NO_BALANCE=1 run drive-pd nobalance "--envs-per-gpu 8192"
run drive-pd balance "--model cassie_tray_box"
CASSIE_NO_BALANCE=1 run drive-pd nobalance "--model cassie_tray_box"
