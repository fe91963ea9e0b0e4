# Build the HIP engine (gfx950) and the CPU oracle.  No cmake, no JIT cache:
# the shared objects are built in-tree so they travel with the gpurun snapshot.
HIPCC      ?= /opt/rocm/bin/hipcc
CC         ?= gcc
ARCH       ?= gfx950

# -ffp-contract=off: the reference (Rust) never fuses a*b+c; hipcc defaults to
# contract=fast.  No fast-math anywhere.  Denormals stay enabled (gfx9 default).
# --offload-compress: the gfx950 code objects are stored zstd-compressed in the fat binary (the HIP runtime of ROCm >= 6.1
# unpacks them at load): libidsp_hip.so 90 MB -> about a third.
HIPFLAGS   := --offload-arch=$(ARCH) --offload-compress -O3 -std=c++17 -fPIC -ffp-contract=off \
              -fno-fast-math -fno-gpu-flush-denormals-to-zero -fwrapv -Wall -Wno-unused-function -Wno-pass-failed -Iinclude
# -Wno-pass-failed silences "loop not unrolled" for the deliberately partially unrolled loops; the kernels whose
# register arrays DEPEND on full unrolling are guarded by `make check-scratch` (tools/check_scratch.py reads every
# kernel's scratch size from the code-object metadata; tests/test_build_scratch.py runs it) instead of by that warning.
ORCFLAGS   := -O3 -std=c11 -fPIC -ffp-contract=off -fno-fast-math \
              -fwrapv -Wall -Wextra -D_GNU_SOURCE

CSRC       := idsp_amd/csrc
# the translation units that compile longest go first, so that `make -j` does not end on one of them
HIP_SLOW   := $(addprefix $(CSRC)/,biquad_f32_df1.hip biquad_f32_df2t.hip normal.hip normal_wdf.hip biquad_i32_dither.hip cic_int_i32_hi.hip biquad_i32_wide.hip lowpass.hip \
              biquad_f64.hip biquad_f64_df2t.hip cascade.hip cascade_f32.hip biquad_bylane_f.hip biquad_bylane_f64.hip biquad_bylane_i32.hip biquad_bylane_i32_wide.hip \
              biquad_i32_df1.hip cic_int_i64_hi.hip lockin_generic.hip lockin_stream_arg.hip cic_dec_i32_hi.hip)
HIP_SRCS   := $(HIP_SLOW) $(filter-out $(HIP_SLOW),$(wildcard $(CSRC)/*.hip))
HIP_OBJS   := $(HIP_SRCS:.hip=.o)
HIP_HDRS   := $(wildcard $(CSRC)/*.h) include/idsp_hip.h

LIB        := idsp_amd/lib/libidsp_hip.so
ORACLE     := oracle/_build/libidsp_oracle.so
# -march=native variant: built ON the machine that times it (bench.py cpu_baseline)
ORACLE_NAT := oracle/_build/libidsp_oracle_native.so

all: $(LIB) $(ORACLE)
lib: $(LIB)
oracle: $(ORACLE)

# No SLP vectorisation anywhere.  (i) It packs two independent i64 MACs into <2 x i64> operations, which the back end
# lowers to generic 64-bit multiplies (v_mul_lo_u32 / v_mad_u64_u32 chains) instead of v_mad_i64_i32 — seen to come and
# go with unrelated edits.  (ii) In the half-band FIR files it produces v_pk_add_f32 / v_pk_mul_f32; round 1 kept it on
# there for that, but packed f32 VALU is no faster than two scalar operations on gfx950 and costs register shuffles:
# every half-band kernel measured 1-4 % faster without it (x4 interpolator 10 %; profiles/r02_exp_hbf_slp.txt).
NOSLP_FLAGS := -fno-slp-vectorize

# Header dependencies per translation unit (-MMD writes csrc/x.d next to csrc/x.o); an object without a .d file yet
# depends on every header.
.SECONDEXPANSION:
$(CSRC)/%.o: $(CSRC)/%.hip $$(if $$(wildcard $(CSRC)/$$*.d),,$(HIP_HDRS))
	$(HIPCC) $(HIPFLAGS) $(NOSLP_FLAGS) -MMD -MP -c $< -o $@
-include $(HIP_OBJS:.o=.d)

$(LIB): $(HIP_OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) --offload-arch=$(ARCH) --offload-compress -shared -fPIC -o $@ $(HIP_OBJS)

$(ORACLE): oracle/idsp_oracle.c oracle/idsp_oracle.h include/idsp_hip.h
	@mkdir -p $(dir $@)
	$(CC) $(ORCFLAGS) -shared -o $@ oracle/idsp_oracle.c -lm -lpthread

oracle-native: $(ORACLE_NAT)
$(ORACLE_NAT): oracle/idsp_oracle.c oracle/idsp_oracle.h include/idsp_hip.h
	@mkdir -p $(dir $@)
	$(CC) $(ORCFLAGS) -march=native -shared -o $@ oracle/idsp_oracle.c -lm -lpthread

# C++ host layer (include/idsp_hip.hpp) test program: plain g++, links the C ABI only
build/test_host: tests/cpp/test_host.cpp include/idsp_hip.hpp include/idsp_hip.h $(LIB)
	@mkdir -p build
	g++ -std=c++17 -O1 -Wall -Iinclude tests/cpp/test_host.cpp -Lidsp_amd/lib -lidsp_hip -Wl,-rpath,'$$ORIGIN/../idsp_amd/lib' -o $@

# coefficient front-end test program: host code only, runs without a GPU
build/test_coeff: tests/cpp/test_coeff.cpp include/idsp_hip.hpp include/idsp_hip.h $(LIB)
	@mkdir -p build
	g++ -std=c++17 -O1 -Wall -Iinclude tests/cpp/test_coeff.cpp -Lidsp_amd/lib -lidsp_hip -Wl,-rpath,'$$ORIGIN/../idsp_amd/lib' -o $@

check-scratch: $(LIB)
	python3 tools/check_scratch.py --lib $(LIB)

# no kernel may reach memory through flat_* instructions (tools/check_flat.py, round 6)
check-flat: $(LIB)
	python3 tools/check_flat.py --lib $(LIB)

clean:
	rm -f $(HIP_OBJS) $(HIP_OBJS:.o=.d) $(LIB) $(ORACLE) $(ORACLE_NAT)

.PHONY: all lib oracle oracle-native clean check-scratch check-flat
