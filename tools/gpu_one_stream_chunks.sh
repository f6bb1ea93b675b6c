for rep in 1 2; do for ck in 4 5 6 7; do
CASSIE_CHUNKS=$ck python bench.py --streams 1 --no-cpu-baseline --no-step-pd --no-other-mode --no-randomised --steps 500 --repeats 6 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('one stream, whole-batch launches, chunks $ck: %.3f M (min %.3f max %.3f) kernel_ms %.3f' % (d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['roofline']['kernel_ms']))"
done; done
