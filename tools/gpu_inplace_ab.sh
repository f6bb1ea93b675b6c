# A/B of the fast kernel's form inside one lease, alternating: plain (kernel + list-walking pass), in place, auto (per range; the default)
B="--no-cpu-baseline --no-step-pd --no-other-mode --no-randomised --steps 500 --repeats 6"
run() { python bench.py $@ $B 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json,sys; d=json.load(open('gpurun_out/t.json')); print('%-50s %7.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f stream_ms %.3f launches plain/in place %s' % (' '.join(sys.argv[1:]), d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['max_qpos_rel_err'], d['roofline']['kernel_ms'], d['roofline']['stream_ms_per_policy_step'], d.get('fast_kernel_launches_plain_in_place')))" "$@"; }
for rep in 1 2; do
for args in "" "--model cassie_hfield" "--target-spread 10" "--model cassie_hfield --target-spread 10"; do
  for form in plain in-place auto; do printf "%-9s" $form; CASSIE_FAST_KERNEL_FORM=$form run $args; done
done; done
for form in plain auto; do
CASSIE_FAST_KERNEL_FORM=$form python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd --no-other-mode --no-randomised 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('short $form', d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d.get('fast_kernel_launches_plain_in_place'))"
done
