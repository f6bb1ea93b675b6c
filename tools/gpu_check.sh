mkdir -p gpurun_out; nproc > gpurun_out/nproc.txt; rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/smi.txt
(time python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.log 2>&1
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
python bench.py --steps 1000 --warmup 100 > gpurun_out/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 50 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT; tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3; cat gpurun_out/bench.log | tail -3
NSUB=50 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50.txt 2>&1
python tools/stage_profile.py 4096 > gpurun_out/stage_profile.txt 2>&1
