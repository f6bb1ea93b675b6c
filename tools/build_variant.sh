#!/bin/bash
# tools/build_variant.sh NAME "EXTRA HIPCC FLAGS" ["LINES OF A PREFIX HEADER"] -- builds the product library with extra
# device-compile flags (and, optionally, a prefix header forced into every HIP translation unit, e.g. a WV_OCC attribute)
# into cassie-mujoco-sim_amd/lib/variants/libcassiemujoco_NAME.so (git-ignored; select it with CASSIE_LIB=<path>).  Used to
# measure compiler-option / source-variant experiments side by side on ONE GPU box (clocks differ from box to box).
set -e
NAME=$1; EXTRA=$2; PREFIX=$3
cd "$(dirname "$0")/.."
mkdir -p build/variants cassie-mujoco-sim_amd/lib/variants
printf '%s\n' "$PREFIX" > build/variants/$NAME.h
OBJS=$(ls build/*.o | grep -v '\.hip\.o')
for src in cassie-mujoco-sim_amd/csrc/*.hip; do
  o=build/variants/$(basename $src .hip)_$NAME.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Icassie-mujoco-sim_amd/csrc -ffp-contract=on ${SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp} ${LICM--mllvm -disable-machine-licm} $EXTRA \
      -include build/variants/$NAME.h -c $src -o $o 2> build/variants/${NAME}_$(basename $src .hip).log &
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cassie-mujoco-sim_amd/lib/variants/libcassiemujoco_$NAME.so $OBJS \
    -Wl,--whole-archive /root/reference/src/libagilitycassie.a -Wl,--no-whole-archive -lm -lpthread
echo "built variant $NAME"
