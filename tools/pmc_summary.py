#!/usr/bin/env python3
"""Reduces the rocprofv3 --pmc CSVs of tools/gpu_pmc_all.sh to one JSON: per-launch averages of every counter over the
launches of the bench's 50-substep step kernel, and the per-env-step figures DESIGN.md quotes.  FETCH_SIZE / WRITE_SIZE
are in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-byte requests at 64 B)."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
acc = collections.defaultdict(list)
kernel = None
for p in sorted(glob.glob(root + "/p*/*/*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(p)) if "cassie_step_kernel" in r.get("Kernel_Name", "")]
    if not rows:
        continue
    # keep the fused launches only (the longest-running dispatches of the step kernel): drop the shortest third
    by_disp = collections.defaultdict(dict)
    for r in rows:
        by_disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        kernel = r["Kernel_Name"]
    for d in by_disp.values():
        for k, v in d.items():
            acc[k].append(v)
per = {}
for k, v in acc.items():
    v = sorted(v)
    v = v[len(v) // 3:]               # the timed region's full 50-substep launches dominate the upper two thirds
    per[k] = sum(v) / len(v)
n_env_steps = 4096 * 50
g = lambda k: per.get(k, float("nan"))
derived = {
    "env_steps_per_launch": n_env_steps,
    "valu_insts_per_env_step": g("SQ_INSTS_VALU") / n_env_steps,
    "salu_insts_per_env_step": g("SQ_INSTS_SALU") / n_env_steps,
    "lds_insts_per_env_step": g("SQ_INSTS_LDS") / n_env_steps,
    "vmem_reads_per_env_step": g("SQ_INSTS_VMEM_RD") / n_env_steps,
    "vmem_writes_per_env_step": g("SQ_INSTS_VMEM_WR") / n_env_steps,
    "wave_cycles_per_env_step": 4 * g("SQ_WAVE_CYCLES") / n_env_steps,
    "frac_wait_any": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
    "frac_valu_active": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
    "frac_wait_lds": g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"),
    "frac_wait_inst_fetch": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
    "lds_bank_conflict_frac_of_lds_active": g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS"),
    "icache_hit": g("SQC_ICACHE_HITS") / g("SQC_ICACHE_REQ"),
    "hbm_read_bytes_per_launch": 2 * 1024 * g("FETCH_SIZE"),
    "hbm_write_bytes_per_launch": 1024 * g("WRITE_SIZE"),
    "note": "SQ_* cycle counters are quad-cycles; FETCH_SIZE/WRITE_SIZE are KB, FETCH_SIZE doubled per MI355X_MICROARCH.md "
            "(gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated",
}
print(json.dumps({"kernel": kernel, "launch": "4096 envs x 50 fused substeps = 204800 env-steps (bench.py --steps 100 --warmup 50)",
                  "per_launch": per, "derived": derived}, indent=1))
