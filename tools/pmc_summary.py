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
# counter values per kernel name: a stepping launch is two kernels since round 3 (the row-capped fast instantiation and the
# full one that only finishes handed-over envs); the summary is of the one that does the work -- the most wave cycles
by_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob(root + "/p*/*/*counter_collection.csv")):
    by_disp = collections.defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(p)):
        if "cassie_step_kernel" not in r.get("Kernel_Name", ""):
            continue
        by_disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for disp, d in by_disp.items():
        for k, v in d.items():
            by_kernel[names[disp]][k].append(v)
if not by_kernel:
    raise SystemExit("no cassie_step_kernel dispatches under " + root)
kernel = max(by_kernel, key=lambda n: sum(by_kernel[n].get("SQ_WAVE_CYCLES", [0.0])))
others = {n: len(next(iter(c.values()))) for n, c in by_kernel.items() if n != kernel}
acc = by_kernel[kernel]
per = {}
for k, v in acc.items():
    v = sorted(v)
    v = v[len(v) // 3:]               # the timed region's full 50-substep launches dominate the upper two thirds
    per[k] = sum(v) / len(v)
n_env_steps = 4096 * 50
g = lambda k: per.get(k, float("nan"))
derived = {
    "env_steps_per_launch": n_env_steps,
    "envs_per_launch": 4096,
    # whole-batch launches of 4096 envs x 50 substeps go as PMC_CHUNKS chunks per env (phys_batch.hip: 7 since round 6; the v36 / v37 files: 4)
    "chunks_per_env_launch": int(__import__("os").environ.get("PMC_CHUNKS", "7")),
    "valu_insts_per_env_step": g("SQ_INSTS_VALU") / n_env_steps,
    "salu_insts_per_env_step": g("SQ_INSTS_SALU") / n_env_steps,
    "lds_insts_per_env_step": g("SQ_INSTS_LDS") / n_env_steps,
    "vmem_reads_per_env_step": g("SQ_INSTS_VMEM_RD") / n_env_steps,
    "vmem_writes_per_env_step": g("SQ_INSTS_VMEM_WR") / n_env_steps,
    "wave_cycles_per_env_step": 4 * g("SQ_WAVE_CYCLES") / n_env_steps,
    "frac_wait_any": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
    "frac_valu_active": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
    "frac_wait_lds": g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"),
    # SQ_WAIT_INST_ANY = wave cycles spent waiting for ANY instruction to issue (issue arbitration between the waves of a SIMD,
    # dependency stalls that are not counted under a specific unit) -- NOT instruction fetch (round 4 mislabelled it so).  What
    # fetch costs is bounded by the instruction-cache misses: a miss is a trip to L2 of a few hundred clocks that a wave at the end
    # of its fetch window waits out
    "frac_wait_issue": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
    "ifetch_requests_per_env_step": g("SQ_IFETCH") / n_env_steps,
    "icache_misses_per_env_step": g("SQC_ICACHE_MISSES") / n_env_steps,
    # upper bound of the fetch stall: every miss waited out in full by one wave at ~500 shader clocks (125 quad-cycles) a miss
    "frac_fetch_stall_upper_bound": 125.0 * g("SQC_ICACHE_MISSES") / g("SQ_WAVE_CYCLES"),
    "lds_bank_conflict_frac_of_lds_active": g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS"),
    "icache_hit": g("SQC_ICACHE_HITS") / g("SQC_ICACHE_REQ"),
    # the dynamic VALU instruction mix by class, per env-step (round 6; SQ_INSTS_VALU_* of gfx950).  "other" = VALU instructions in
    # none of the counted classes: moves, v_cndmask, compares, v_readlane / v_writelane (cross-lane reads, SGPR spill traffic), DPP moves
    "valu_mix_per_env_step": {k: g("SQ_INSTS_VALU_" + k) / n_env_steps for k in
                              ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "MFMA_F64", "INT32", "INT64", "CVT", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32")},
    "valu_other_per_env_step": (g("SQ_INSTS_VALU") - sum(g("SQ_INSTS_VALU_" + k) for k in
                                ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "MFMA_F64", "INT32", "INT64", "CVT", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32"))) / n_env_steps,
    "branches_per_env_step": g("SQ_INSTS_BRANCH") / n_env_steps,
    "lds_loads_per_env_step": g("SQ_INSTS_LDS_LOAD") / n_env_steps,
    "lds_stores_per_env_step": g("SQ_INSTS_LDS_STORE") / n_env_steps,
    "hbm_read_bytes_per_launch": 2 * 1024 * g("FETCH_SIZE"),
    "hbm_write_bytes_per_launch": 1024 * g("WRITE_SIZE"),
    "note": "SQ_* cycle counters are quad-cycles; FETCH_SIZE/WRITE_SIZE are KB, FETCH_SIZE doubled per MI355X_MICROARCH.md "
            "(gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated",
}
print(json.dumps({"kernel": kernel, "other_step_kernels_in_the_run": others, "launch": "4096 envs x 50 fused substeps = 204800 env-steps (bench.py --streams 1 --steps 100 --warmup 50: whole-batch launches, one at a time)",
                  "per_launch": per, "derived": derived}, indent=1))
