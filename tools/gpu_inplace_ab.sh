B="--no-cpu-baseline --no-step-pd --no-other-mode --steps 500 --repeats 6"
run() { python bench.py $@ $B 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json,sys; d=json.load(open('gpurun_out/t.json')); print('%-60s %7.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f stream_ms %.3f handed %.4f' % (' '.join(sys.argv[1:]), d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['max_qpos_rel_err'], d['roofline']['kernel_ms'], d['roofline']['stream_ms_per_policy_step'], d.get('frac_envs_handed_over_to_the_full_kernel_in_the_last_launch') or 0))" "$@"; }
for rep in 1 2; do
for args in "" "--target-spread 10" "--model cassie_hfield --target-spread 10"; do
  echo -n "inplace    "; run $args
  echo -n "NO_INPLACE "; CASSIE_NO_INPLACE=1 run $args
done; done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd --no-other-mode 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('short inplace', d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6)"
CASSIE_NO_INPLACE=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd --no-other-mode 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('short NO_INPLACE', d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6)"
