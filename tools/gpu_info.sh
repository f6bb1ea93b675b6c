mkdir -p gpurun_out
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; taskset -p $$; lscpu | head -20; python -c "
import sys; sys.path.insert(0,'cassie-mujoco-sim_amd')
from cassie_amd._lib import lib
print('usable', lib().cassie_host_cpu_count())") > gpurun_out/cpuinfo.txt 2>&1
cat gpurun_out/cpuinfo.txt
