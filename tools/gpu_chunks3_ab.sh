for rep in 1 2 3; do for ck in 2 3; do
CASSIE_CHUNKS=$ck python bench.py --no-cpu-baseline --no-step-pd --no-other-mode --no-randomised --steps 500 --repeats 6 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('long,  chunks $ck: %.3f M (min %.3f max %.3f) kernel_ms %.3f' % (d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['roofline']['kernel_ms']))"
CASSIE_CHUNKS=$ck python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd --no-other-mode --no-randomised 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('short, chunks $ck: %.3f M (min %.3f max %.3f) kernel_ms %.3f' % (d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['roofline']['kernel_ms']))"
done; done
