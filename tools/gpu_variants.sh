# A/B of library variants (tools/build_variant.sh) on ONE box, alternating, twice: VARIANTS="base asplit ..." (the in-tree
# library is "tree"); prints the bench line's value per run.  Box clocks first (they differ from box to box by up to 30 %).
mkdir -p gpurun_out
(rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -E "sclk|mclk|Power|Perf" | head -8) > gpurun_out/box_clocks.txt; cat gpurun_out/box_clocks.txt
for rep in 1 2; do
for v in ${VARIANTS:-tree}; do
  if [ $v = tree ]; then unset CASSIE_LIB; else export CASSIE_LIB=$PWD/cassie-mujoco-sim_amd/lib/variants/libcassiemujoco_$v.so; fi
  timeout 300 python bench.py --model ${MODEL:-cassie} --steps ${STEPS:-500} --warmup 50 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/variant_$v.err | grep '^{"metric"' > gpurun_out/variant_${v}_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/variant_${v}_$rep.json"))
print("%-10s run $rep: %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f" % ("$v", d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"]))
PY
done
done
unset CASSIE_LIB
