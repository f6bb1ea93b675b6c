# A/B inside one lease: how long an env stays in the 63-row code of the in-place fast kernel (CASSIE_INPLACE_STAY_ROWS: 0 = one substep)
B="--no-cpu-baseline --no-step-pd --no-other-mode --no-randomised --steps 500 --repeats 4"
run() { python bench.py $@ $B 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json,sys; d=json.load(open('gpurun_out/t.json')); print('%-52s %7.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f stream_ms %.3f forms %s' % (' '.join(sys.argv[1:]), d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['max_qpos_rel_err'], d['roofline']['kernel_ms'], d['roofline']['stream_ms_per_policy_step'], d.get('fast_kernel_launches_plain_in_place')))" "$@"; }
for rep in 1 2; do
for args in "--model cassie_hfield --hfield-contacts prism" "--target-spread 10" "--model cassie_hfield --target-spread 10"; do
  for stay in 0 27 31 20; do printf "stay %-3s" $stay; CASSIE_INPLACE_STAY_ROWS=$stay run $args; done
done; done
