# the launch-order kernel behind every launch of more than N substeps (CASSIE_ORDER_EVERY=N; 7 = round 5's rule, 25 = the default now), the driver's command
for rep in 1 2 3; do for n in 7 25; do
CASSIE_ORDER_EVERY=$n python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd --no-other-mode --no-randomised 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('short, order kernel behind launches of more than $n substeps: %.3f M (min %.3f max %.3f)' % (d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6))"
done; done
