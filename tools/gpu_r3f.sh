# PMC passes of the cassie and cassie_hfield kernels with the two-kernel-aware summary (tools/pmc_summary.py)
for m in cassie cassie_hfield; do
  rm -rf gpurun_out/pmc; MODEL=$m bash tools/gpu_pmc_all.sh > gpurun_out/pmc_all_$m.log 2>&1; cp gpurun_out/pmc_summary.json gpurun_out/pmc_summary_$m.json
done
python - <<'PY'
import json
for m in ("cassie", "cassie_hfield"):
    j = json.load(open("gpurun_out/pmc_summary_%s.json" % m)); d = j["derived"]
    print(m, j["kernel"], j["other_step_kernels_in_the_run"], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k != "note"})
PY
