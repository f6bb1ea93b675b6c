#!/bin/bash
# tools/build_variant.sh NAME "EXTRA HIPCC FLAGS" -- builds the product library with extra device-compile flags into
# cassie-mujoco-sim_amd/lib/variants/libcassiemujoco_NAME.so (git-ignored; select it with CASSIE_LIB=<path>).  Used to
# measure compiler-option / source-variant experiments side by side on one GPU box (tools/gpu_variants.sh).
set -e
NAME=$1; EXTRA=$2
cd "$(dirname "$0")/.."
mkdir -p build/variants cassie-mujoco-sim_amd/lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Icassie-mujoco-sim_amd/csrc -ffp-contract=on ${SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp} $EXTRA \
    -c cassie-mujoco-sim_amd/csrc/phys_batch.hip -o build/variants/phys_batch_$NAME.o 2> build/variants/$NAME.log
OBJS=$(ls build/*.o | grep -v phys_batch)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cassie-mujoco-sim_amd/lib/variants/libcassiemujoco_$NAME.so $OBJS build/variants/phys_batch_$NAME.o \
    -Wl,--whole-archive /root/reference/src/libagilitycassie.a -Wl,--no-whole-archive -lm -lpthread
echo "built variant $NAME"
