# A/B inside one lease, library variants (VARIANTS="tree prev"): config 4 with the default and with the prism contact definition
B="--no-cpu-baseline --no-step-pd --no-other-mode --no-randomised --steps 500 --repeats 4 --model cassie_hfield"
for rep in 1 2; do for def in default prism; do for v in ${VARIANTS:-tree prev}; do
if [ $v = tree ]; then unset CASSIE_LIB; else export CASSIE_LIB=$PWD/cassie-mujoco-sim_amd/lib/variants/libcassiemujoco_$v.so; fi
python bench.py $B --hfield-contacts $def 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json,sys; d=json.load(open('gpurun_out/t.json')); print('%-8s %-6s %7.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f stream_ms %.3f forms %s' % ('$def', '$v', d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['max_qpos_rel_err'], d['roofline']['kernel_ms'], d['roofline']['stream_ms_per_policy_step'], d.get('fast_kernel_launches_plain_in_place')))"
done; done; done
unset CASSIE_LIB
