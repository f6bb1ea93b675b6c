# round 4, second lease: where does the two-wave form lose the difference between its substep latency (x 0.78) and its rate (x 1.11)?
# (a) the resume pass (2048 one-wave workgroups of the full kernel, each needing an EMPTY SIMD and 40 KB of LDS) -- skipped through the
#     measurement knob on a workload that hands nothing over; (b) the stamps of a launch that shares the GPU with another batch's;
# (c) SQ counters of both forms on this box.
mkdir -p gpurun_out
ab() { # waves skip
  for rep in 1 2; do
  env CASSIE_WAVES_PER_ENV=$1 ${2:+CASSIE_DEBUG_SKIP_RESUME_PASS=1} timeout 300 python bench.py --steps 500 --warmup 50 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4b.err | grep '^{"metric"' > gpurun_out/r4b_w$1_skip${2:-0}_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4b_w$1_skip${2:-0}_$rep.json"))
print("waves=$1 skip_resume=${2:-0} run $rep: %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f stream_ms %.3f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step", 0)))
PY
  done
}
(ab 2; ab 2 1; ab 1; ab 1 1) 2>&1 | tee gpurun_out/resume_pass_ab.txt
for w in 2 1; do NSUB=50 WAVES=$w TWO_STREAM_LOAD=1 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_loaded_w$w.txt 2>&1; head -30 gpurun_out/stage_profile_loaded_w$w.txt; done
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for w in 2 1; do
  rm -rf $R/gpurun_out/pmc; mkdir -p $R/gpurun_out/pmc
  CMD="python $R/bench.py --mode drive-pd --streams 1 --steps 100 --warmup 50 --no-cpu-baseline --no-step-pd --no-other-mode --parity-envs 4"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    CASSIE_WAVES_PER_ENV=$w timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc/p$i -- $CMD > $R/gpurun_out/pmc/p$i.log 2>&1
  done
  (cd $R; python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc_sq_w$w.json; python - <<PY
import json
d = json.load(open("gpurun_out/pmc_sq_w$w.json")); print("waves=$w", d["kernel"][-40:], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["derived"].items() if k != "note" and v == v})
PY
)
done
