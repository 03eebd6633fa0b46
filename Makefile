# Build without Python: what a Rust / C host's build script (build.rs, cmake ExternalProject) would call.
#   make lib       valida_amd/libvgpu.so   hipcc --offload-arch=gfx950, the same sources and flags as valida_amd/build.py
#   make c_host    examples/c_host         a C99 host linked against it
#   make oracle    oracle/liboracle.so     the CPU checker (test infrastructure only)
#   make test      the CPU test suite
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := valida_amd/csrc
SRCS := kernels/ntt.hip kernels/layout.hip kernels/merkle.hip kernels/poseidon_mmcs.hip kernels/perm.hip kernels/quotient.hip kernels/open.hip \
        kernels/tracegen.hip host/prover.cpp host/sharded_prover.cpp capi.cpp
OBJS := $(addprefix build/make/,$(addsuffix .o,$(subst /,_,$(SRCS))))
HDRS := $(shell find $(CSRC) -name '*.hpp' -o -name '*.h') include/vgpu.h
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-unused-result

.PHONY: lib c_host oracle test clean
lib: valida_amd/libvgpu.so
define RULE
build/make/$(subst /,_,$(1)).o: $(CSRC)/$(1) $(HDRS)
	@mkdir -p build/make
	$(HIPCC) $(FLAGS) -x hip -c $$< -o $$@
endef
$(foreach s,$(SRCS),$(eval $(call RULE,$(s))))
valida_amd/libvgpu.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl
c_host: lib
	gcc -std=c99 -pedantic -Wall -Wextra -Werror -Iinclude examples/c_host.c -o examples/c_host -Lvalida_amd -lvgpu -Wl,-rpath,$(CURDIR)/valida_amd
oracle:
	$(MAKE) -C oracle
test: lib oracle
	python -m pytest tests -q -m "not gpu"
clean:
	rm -rf build/make examples/c_host
