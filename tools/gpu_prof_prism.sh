R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_prism -- python $R/bench.py --model cassie_hfield --hfield-contacts prism --no-cpu-baseline --no-step-pd --no-other-mode --steps 500 --repeats 3 > $R/gpurun_out/prof_prism.log 2>&1
cd $R; f=$(ls -t gpurun_out/prof_prism/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_prism.csv; head -8 $f | cut -c1-260
