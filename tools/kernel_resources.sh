#!/bin/bash
# Compile only the headline instantiation (NVP=32, static Cassie topology) and print its register / scratch / LDS use.
set -e
T=$(mktemp -d)
cat > $T/one.hip <<EOS
#include <hip/hip_runtime.h>
#include "wave.h"
#include "physics_kernel.h"
#include "topo_static.h"
template __global__ void ck::cassie_step_kernel<32, ck::TopoCassie32, ${FEAT:-0}>(ck::PhysIO);
EOS
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --offload-device-only -Iinclude -Icassie-mujoco-sim_amd/csrc \
  -ffp-contract=on ${SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp} $EXTRA -Rpass-analysis=kernel-resource-usage ${KEEP:+-save-temps=obj} -c $T/one.hip -o $T/one.o 2>&1 |
  grep -E "VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
[ -n "$KEEP" ] && cp $T/one-hip-amdgcn-amd-amdhsa-gfx950.s "$KEEP" || true
rm -rf $T
