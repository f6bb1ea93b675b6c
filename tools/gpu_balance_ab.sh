for rep in 1 2 3; do for nb in 0 1; do
if [ $nb = 1 ]; then export CASSIE_NO_BALANCE=1; else unset CASSIE_NO_BALANCE; fi
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd --no-other-mode --no-randomised 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('short, no-balance $nb: %.3f M (min %.3f max %.3f)' % (d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6))"
done; done
unset CASSIE_NO_BALANCE
for nb in 0 1; do
if [ $nb = 1 ]; then export CASSIE_NO_BALANCE=1; else unset CASSIE_NO_BALANCE; fi
python bench.py --no-cpu-baseline --no-step-pd --no-other-mode --no-randomised --steps 500 --repeats 6 2>/dev/null | grep '^{' > gpurun_out/t.json; python -c "
import json; d=json.load(open('gpurun_out/t.json')); print('long,  no-balance $nb: %.3f M' % (d['value']/1e6))"
done
