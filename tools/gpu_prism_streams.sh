B="--no-cpu-baseline --no-step-pd --no-other-mode --no-randomised --steps 500 --repeats 4 --model cassie_hfield --hfield-contacts prism"
for st in 1 2 4 8; do for form in plain auto; do
CASSIE_FAST_KERNEL_FORM=$form python bench.py $B --streams $st 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json,sys; d=json.load(open('gpurun_out/t.json')); print('streams $st $form %7.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f stream_ms %.3f forms %s wide %s handed %s' % (d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['max_qpos_rel_err'], d['roofline']['kernel_ms'], d['roofline']['stream_ms_per_policy_step'], d.get('fast_kernel_launches_plain_in_place'), d.get('frac_envs_in_the_127_row_pass_in_the_last_launch'), d.get('frac_envs_handed_over_to_the_full_kernel_in_the_last_launch')))"
done; done
