#!/bin/bash
# Compiles the headline instantiations of the step kernel (NVP = 32, static Cassie topology: the row-capped fast one, MAXR = 31,
# that every stepping launch starts, and the full one, MAXR = 63, that finishes handed-over envs) and prints their register /
# scratch / LDS use.  FEAT=1 for the height-field instantiations; NVP=40 TOPO=TopoCassieTray38 FEAT=2 MAXRS=63 for the tray
# kernel; KEEP=file keeps the ISA of the last one (tools/asm_segments.py, tools/asm_liveness.py read it).
set -e
cd "$(dirname "$0")/.."
for MAXR in ${MAXRS:-31 63}; do
T=$(mktemp -d)
cat > $T/one.hip <<EOS
#include <hip/hip_runtime.h>
#include "wave.h"
#include "physics_kernel.h"
#include "topo_static.h"
template __global__ void ck::cassie_step_kernel<${NVP:-32}, ck::${TOPO:-TopoCassie32}, ${FEAT:-0}, $MAXR, ${NW:-1}, ${WALK:-false}, ${WPS:-${NW:-1}}${INROWS:+, $INROWS}>(ck::PhysIO);
EOS
echo "cassie_step_kernel<${NVP:-32}, ${TOPO:-TopoCassie32}, FEAT=${FEAT:-0}, MAXR=$MAXR, NW=${NW:-1}, WALK=${WALK:-false}, WPS=${WPS:-${NW:-1}}${INROWS:+, INROWS=$INROWS}>:"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --offload-device-only -Iinclude -Icassie-mujoco-sim_amd/csrc \
  -ffp-contract=on ${SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp} ${LICM--mllvm -disable-machine-licm} $EXTRA -Rpass-analysis=kernel-resource-usage ${KEEP:+-save-temps=obj} -c $T/one.hip -o $T/one.o 2>&1 |
  grep -E "VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tail -8 | sed 's/^/    /'
[ -n "$KEEP" ] && cp $T/one-hip-amdgcn-amd-amdhsa-gfx950.s "$KEEP" || true
rm -rf $T
done
