# Sanitizer build of the HOST code of libzkhip (CPU only; tools/asan_cpu.sh drives it: make -f asan.mk).  Kept out of the Makefile and
# listed in .gpurunignore: GPU sanitizer builds are not available on the pool, and nothing on the GPU box needs this file.
include Makefile
SAN = -fsanitize=address
ASAN_OUT ?= ../libzkhip_asan.so
HOST_SRCS = $(filter %.cpp,$(SRCS))
# .hip files with host-side verifiers and witness builders: device code as shipped, host code sanitized (-Xarch_host)
HOSTY_HIP = fri_chip.hip sha256_chip.hip hal.hip
ASAN_OBJS = $(patsubst %,build/asan/%.o,$(HOST_SRCS) $(HOSTY_HIP)) $(patsubst %,build/%.o,$(filter-out $(HOSTY_HIP),$(filter %.hip,$(SRCS))))
asan: $(ASAN_OUT)
$(ASAN_OUT): $(ASAN_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(SAN) -o $@ $(ASAN_OBJS)
build/asan/p2_x16.cpp.o: p2_x16.cpp $(HDRS)
	@mkdir -p build/asan
	$(HIPCC) -O1 -g -std=c++17 -fPIC -Wall -mavx512f -mavx512dq $(SAN) -fno-omit-frame-pointer -x c++ -c $< -o $@
build/asan/%.cpp.o: %.cpp $(HDRS)
	@mkdir -p build/asan
	$(HIPCC) -O1 -g -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Wno-option-ignored $(SAN) -fno-omit-frame-pointer -x hip -c $< -o $@
build/asan/%.hip.o: %.hip $(HDRS)
	@mkdir -p build/asan
	$(HIPCC) -O1 -g -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Wno-option-ignored -Xarch_host $(SAN) -fno-omit-frame-pointer -c $< -o $@
.PHONY: asan
