# the driver's command (20-step fenced regions) against the number of env ranges (streams) and the chunk count
for rep in 1 2; do for st in 1 2 3 4; do for ck in 2 4 6; do
CASSIE_CHUNKS=$ck python bench.py --gpus 1 --steps 20 --warmup 5 --streams $st --no-cpu-baseline --no-step-pd --no-other-mode --no-randomised 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('short, streams $st chunks $ck: %.3f M (min %.3f max %.3f) kernel_ms %.3f' % (d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['roofline']['kernel_ms']))"
done; done; done
