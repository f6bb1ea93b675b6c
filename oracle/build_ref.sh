#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE.  Builds oracle/_ref/libref_hostpath.so from the reference's
# OWN code where it lies under /root/reference (nothing is copied into the repository; oracle/_ref/ is
# git-ignored):
#   * the encoder / motor models are taken verbatim from src/cassiemujoco.c by pattern (the filter
#     constants and types, the NUM_* defines, and the functions drive_encoder / joint_encoder / motor),
#     written to oracle/_ref/ref_extract.inc at build time and compiled inside oracle/ref_hostpath_harness.c,
#     which supplies the few mjModel / mjData fields those functions touch;
#   * the Agility blocks come from src/libagilitycassie.a (closed binary, whole-archived).
# The full reference library cannot be built: it needs MuJoCo 2.1.0 headers and binaries.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$REF/src/cassiemujoco.c
[ -f "$SRC" ] || { echo "reference not present at $REF: nothing to build"; exit 0; }
mkdir -p "$HERE/_ref"
{
  sed -n '/^#define DRIVE_FILTER_NB/,/^} joint_filter_t;/p' "$SRC"
  grep -E '^#define (NUM_DRIVES|NUM_JOINTS|TORQUE_DELAY_CYCLES) ' "$SRC"
  sed -n '/^static void drive_encoder/,/^static void window_close_callback/p' "$SRC" | sed '$d'
} > "$HERE/_ref/ref_extract.inc"
gcc -O2 -std=gnu11 -fPIC -shared -I"$HERE/../include" -I"$HERE/_ref" "$HERE/ref_hostpath_harness.c" \
    -Wl,--whole-archive "$REF/src/libagilitycassie.a" -Wl,--no-whole-archive -lm -o "$HERE/_ref/libref_hostpath.so"
echo "built $HERE/_ref/libref_hostpath.so"
# The reference's own Python wrapper and MJCF files, staged (not committed: oracle/_ref/ is git-ignored, but it travels
# to the GPU box with the snapshot) so that the -m gpu suite can run the UNMODIFIED example/cassiemujoco.py against the
# product library (tests/test_dropin_gpu.py) and so that tests/mujoco_ref.py has the XML a genuine MuJoCo would load.
mkdir -p "$HERE/_ref/example" "$HERE/_ref/model"
cp "$REF/example/cassiemujoco.py" "$REF/example/cassiemujoco_ctypes.py" "$HERE/_ref/example/"
cp "$REF/model/cassie.xml" "$REF/model/cassie_hfield.xml" "$REF/model/cassie_tray_box.xml" "$HERE/_ref/model/"
cp -r "$REF/model/cassie-stl-meshes" "$HERE/_ref/model/"
[ -d "$REF/model/terrains" ] && cp -r "$REF/model/terrains" "$HERE/_ref/model/" || true
echo "staged $HERE/_ref/example and $HERE/_ref/model"
# The reference's own C example programs (example/*.c: the UDP simulator and controller, and the visualiser demos),
# compiled UNMODIFIED from where they lie against this repository's include/ and product library -- the source-level
# drop-in check for C users (tests/test_ref_programs.py).  Needs the product library (make product) to link.
PROD="$HERE/../cassie-mujoco-sim_amd/lib"
if [ -f "$PROD/libcassiemujoco.so" ]; then
  mkdir -p "$HERE/_ref/bin"
  for f in cassiesim cassiectrl cassietest cassievideo test_doublevis test_heelforce test_hfield test_terrain; do
    gcc -O1 -w -std=gnu11 -I"$HERE/../include" "$REF/example/$f.c" -o "$HERE/_ref/bin/$f" \
        -L"$PROD" -lcassiemujoco -Wl,-rpath,'$ORIGIN/../../../cassie-mujoco-sim_amd/lib' -lm -lpthread
  done
  echo "built the reference's example programs into $HERE/_ref/bin"
fi
