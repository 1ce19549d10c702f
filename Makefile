# The native libraries without Python (what a Rust build.rs or a C user runs; `python -c "import __graft_entry__ as g; g.build()"` issues the same commands
# through zkp_amd/build.py and also builds the test-hook library and the oracle).  hipcc cross-compiles gfx950 without a GPU.
CSRC := zkp_amd/csrc
HIP_DEPS := $(wildcard $(CSRC)/*.hip $(CSRC)/*.h include/*.h)
HOST_SRCS := $(sort $(wildcard $(CSRC)/host/*.cpp))
HOST_DEPS := $(HOST_SRCS) $(wildcard $(CSRC)/host/*.h $(CSRC)/host/*.hpp $(CSRC)/host/*.map include/*.h) $(CSRC)/ge25519.h $(CSRC)/fe25519.h $(CSRC)/fe_constants.h

all: zkp_amd/libzkp_mi355x.so zkp_amd/libzkp_toolbox.so

zkp_amd/libzkp_mi355x.so: $(HIP_DEPS)
	hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value $(CSRC)/zkp_kernels.hip -o $@

zkp_amd/libzkp_toolbox.so: $(HOST_DEPS) zkp_amd/libzkp_mi355x.so
	g++ -O3 -std=c++17 -shared -fPIC -pthread -Wall -Wno-unknown-pragmas -I include $(HOST_SRCS) -o $@ -L zkp_amd -lzkp_mi355x -Wl,-rpath,'$$ORIGIN' -Wl,--version-script=$(CSRC)/host/exports.map

# the reference's DLEQ test in C99 against the two libraries alone: `build/dleq_c_abi` (host backend), `build/dleq_c_abi gpu 4096`
example: all
	mkdir -p build
	gcc -std=c99 -Wall -Wextra -pedantic -I include examples/dleq_c_abi.c -L zkp_amd -lzkp_toolbox -lzkp_mi355x -Wl,-rpath,$(CURDIR)/zkp_amd -o build/dleq_c_abi

clean:
	rm -rf build zkp_amd/libzkp_mi355x.so zkp_amd/libzkp_toolbox.so

.PHONY: all example clean
