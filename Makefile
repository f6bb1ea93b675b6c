# Build of the MI355X-native Cassie physics library and its test infrastructure.
#
#   make            product library  cassie-mujoco-sim_amd/lib/libcassiemujoco.so   (hipcc, gfx950)
#   make oracle     CPU oracle       oracle/libcassie_oracle.so                     (gcc, test-only)
#   make emu        wave emulator    tests/emu/libcassie_emu.so                     (g++, test-only)
#   make models     models/*.cmodel from the reference MJCF (needs /root/reference)
#
# The Agility blocks (pd_input / cassie_core_sim / state_output + pack/unpack) exist
# only as the closed static library shipped with the reference
# (src/libagilitycassie.a); it is whole-archived into the product .so exactly like
# the reference's own Makefile does (reference Makefile:18).  On a box without
# /root/reference the prebuilt .so is used as is.

ROCM      ?= /opt/rocm
HIPCC     ?= $(ROCM)/bin/hipcc
ARCH      ?= gfx950
REF       ?= /root/reference
AGILITY   ?= $(REF)/src/libagilitycassie.a

PKG   := cassie-mujoco-sim_amd
CSRC  := $(PKG)/csrc
LIBD  := $(PKG)/lib
OBJD  := build

CXXFLAGS := -O2 -std=c++17 -fPIC -Iinclude -I$(CSRC)
CFLAGS   := -O2 -std=gnu11 -fPIC -Iinclude -I$(CSRC)
# -amdgpu-sched-strategy=iterative-ilp: the step kernel is one long latency-bound instruction stream at one wave per SIMD;
# scheduling for ILP instead of for occupancy measured +1.8 % (exact-pd) / +2.7 % (drive-pd), profiles/round2/README.md
# -disable-machine-licm: the substep loop is one 20 000-instruction body at the register limit; machine-level LICM hoists the
# materialisation of every fp64 literal and every loop-invariant lane predicate out of it -- into registers the loop does not
# have, i.e. into scratch and SGPR-spill lanes that the loop then reloads.  Without it: scratch 136 -> 0 B, 474 -> 392 SGPR
# spills, and +1 % (profiles/round3/README.md)
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -I$(CSRC) -ffp-contract=on -mllvm -amdgpu-sched-strategy=iterative-ilp \
            -mllvm -disable-machine-licm

HOST_CPP := $(CSRC)/mjcf_loader.cpp $(CSRC)/phys_host.cpp
HOST_C   := $(wildcard $(CSRC)/*.c)
HIP_SRC  := $(wildcard $(CSRC)/*.hip)   # phys_batch.hip + one kernels_*.hip per model family (they compile side by side)
OBJS     := $(patsubst $(CSRC)/%.cpp,$(OBJD)/%.o,$(HOST_CPP)) $(patsubst $(CSRC)/%.c,$(OBJD)/%.o,$(HOST_C)) \
            $(patsubst $(CSRC)/%.hip,$(OBJD)/%.hip.o,$(HIP_SRC))

PRODUCT := $(LIBD)/libcassiemujoco.so

ifneq ($(wildcard $(AGILITY)),)
AGILITY_LINK := -Wl,--whole-archive $(AGILITY) -Wl,--no-whole-archive
else
AGILITY_LINK :=
endif

.PHONY: all product oracle emu models clean apps
all: product oracle emu apps
apps: $(PKG)/bin/cassiesim
product: $(PRODUCT)
oracle: oracle/libcassie_oracle.so
emu: tests/emu/libcassie_emu.so

$(OBJD)/%.o: $(CSRC)/%.cpp $(wildcard $(CSRC)/*.h) $(wildcard include/*.h)
	@mkdir -p $(OBJD)
	g++ $(CXXFLAGS) -c $< -o $@
$(OBJD)/%.o: $(CSRC)/%.c $(wildcard $(CSRC)/*.h) $(wildcard include/*.h)
	@mkdir -p $(OBJD)
	gcc $(CFLAGS) -c $< -o $@
$(OBJD)/%.hip.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.inc) $(wildcard include/*.h)
	@mkdir -p $(OBJD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(PRODUCT): $(OBJS)
	@mkdir -p $(LIBD)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) $(AGILITY_LINK) -lm -lpthread

# the UDP lock-step server (reference example/cassiesim.c role) against the product library
$(PKG)/bin/cassiesim: $(PKG)/apps/cassiesim.c $(PRODUCT)
	@mkdir -p $(PKG)/bin
	gcc -O2 -std=gnu11 -Iinclude -I$(CSRC) $< -o $@ -L$(LIBD) -lcassiemujoco -Wl,-rpath,'$$ORIGIN/../lib' -lm

oracle/libcassie_oracle.so: oracle/cassie_oracle.c oracle/cassie_oracle.h $(CSRC)/cm_model.h
	gcc -O2 -std=gnu11 -fPIC -shared -fopenmp -I$(CSRC) -Ioracle oracle/cassie_oracle.c -o $@ -lm

tests/emu/libcassie_emu.so: tests/emu/emu_runtime.cpp tests/emu/wave.h $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.inc)
	g++ -O2 -std=c++17 -fPIC -shared -Wl,-Bsymbolic -Itests/emu -I$(CSRC) tests/emu/emu_runtime.cpp -o $@

models: product
	python3 tools/make_models.py $(REF)/model models tests/golden

clean:
	rm -rf $(OBJD) $(PRODUCT) oracle/libcassie_oracle.so tests/emu/libcassie_emu.so
