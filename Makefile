# Builds the MI355X-native render library and its checker.
#   make            -> raytracers_amd/libray_mi355x.so  (HIP, gfx950) + tools
#   make oracle     -> oracle/build/liboracle.so        (CPU oracle, test infrastructure)
# -ffp-contract=off is load-bearing on both host and device: parity is bit-exact fp32.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := raytracers_amd/csrc
OBJ     := build/obj
LIB     := raytracers_amd/libray_mi355x.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-parameter
HOSTFLAGS := -O2 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra

all: $(LIB) tools oracle

$(OBJ)/render_kernels.o: $(CSRC)/render_kernels.hip $(CSRC)/lane_core.h $(CSRC)/rt_device.hpp $(CSRC)/treelet.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/bvh_build.o: $(CSRC)/bvh_build.hip $(CSRC)/rt_device.hpp $(CSRC)/lane_core.h $(CSRC)/treelet.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/api.o: $(CSRC)/api.cpp $(CSRC)/rt_internal.hpp $(CSRC)/rt_device.hpp $(CSRC)/rt_host.hpp $(CSRC)/lane_core.h include/ray.h include/rt_mi355x.h $(CSRC)/treelet.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/multi_gpu.o: $(CSRC)/multi_gpu.cpp $(CSRC)/rt_internal.hpp $(CSRC)/rt_device.hpp $(CSRC)/rt_host.hpp include/rt_mi355x.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/host_build.o: $(CSRC)/host_build.cpp $(CSRC)/rt_host.hpp $(CSRC)/treelet.h
	@mkdir -p $(OBJ)
	$(CXX) $(HOSTFLAGS) -c $< -o $@

# (librccl is NOT linked: multi_gpu.cpp loads it on demand, so a single-GPU host needs no RCCL)
$(LIB): $(OBJ)/render_kernels.o $(OBJ)/bvh_build.o $(OBJ)/api.o $(OBJ)/multi_gpu.o $(OBJ)/host_build.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -ldl

# native harness (our own bench front-end; the reference's futhark/main.c links the same way)
build/rtbench: tools/rtbench.c include/ray.h include/rt_mi355x.h $(LIB)
	@mkdir -p build
	$(CC) -O2 -std=gnu99 -Wall -Iinclude -o $@ tools/rtbench.c -Lraytracers_amd -lray_mi355x -Wl,-rpath,'$$ORIGIN/../raytracers_amd' -lm

build/wavesim: tools/wavesim.cpp $(CSRC)/lane_core.h $(CSRC)/rt_host.hpp $(OBJ)/host_build.o
	@mkdir -p build
	$(CXX) $(HOSTFLAGS) -fopenmp -I$(CSRC) -o $@ tools/wavesim.cpp $(OBJ)/host_build.o

# microbenchmark behind the bench line's measured peaks (tools/gpu_issue_peak.sh runs it)
build/issue_peak: tools/issue_peak.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-inline-asm -o $@ $<

build/hip_touch: tools/hip_touch.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O2 -o $@ $<

# the pooled kernel's tile-queue protocol (rt_device.hpp) played on the CPU; in the CPU test suite
build/queue_check: tools/queue_check.cpp $(CSRC)/rt_device.hpp $(CSRC)/lane_core.h
	@mkdir -p build
	$(HIPCC) -O2 -std=c++17 -Wall -I$(CSRC) -o $@ $<

# the DONATE instantiation's mailbox protocol (render_kernels.hip) as a model, random interleavings; in the CPU test suite
build/donate_check: tools/donate_check.cpp
	@mkdir -p build
	$(CXX) -O2 -std=c++17 -Wall -Wextra -o $@ $<

# design tool + CPU check of the treelet numbering / masks (treelet.h) against a plain depth-first walk; in the CPU test suite
build/treelet_probe: tools/treelet_probe.cpp $(CSRC)/lane_core.h $(CSRC)/treelet.h $(CSRC)/rt_host.hpp $(OBJ)/host_build.o
	@mkdir -p build
	$(CXX) $(HOSTFLAGS) -I$(CSRC) -o $@ tools/treelet_probe.cpp $(OBJ)/host_build.o

# the inequalities behind the CULL instantiations (lane_core.h: cull_limit) hammered with the product's binary32 code against __float128; in the CPU test suite
build/cull_bound_check: tools/cull_bound_check.cpp $(CSRC)/lane_core.h $(CSRC)/rt_host.hpp $(OBJ)/host_build.o
	@mkdir -p build
	$(CXX) $(HOSTFLAGS) -fopenmp -I$(CSRC) -o $@ tools/cull_bound_check.cpp $(OBJ)/host_build.o -lquadmath

# the pooled kernel's loop on emulated lanes with the culling rule: tests saved, pixels kept; in the CPU test suite
build/cull_pooled: tools/cull_pooled.cpp $(CSRC)/lane_core.h $(CSRC)/rt_host.hpp $(OBJ)/host_build.o
	@mkdir -p build
	$(CXX) $(HOSTFLAGS) -I$(CSRC) -o $@ tools/cull_pooled.cpp $(OBJ)/host_build.o

# host threads sharing one context and one prepared scene (the context lock): the plain build is in the GPU test suite ...
build/ctx_threads: tools/ctx_threads.cpp include/ray.h include/rt_mi355x.h $(LIB)
	@mkdir -p build
	$(CXX) -O2 -std=c++17 -Wall -Iinclude -pthread -o $@ tools/ctx_threads.cpp -Lraytracers_amd -lray_mi355x -Wl,-rpath,'$$ORIGIN/../raytracers_amd'

# ... and the same program over the library's HOST code built with -fsanitize=thread (device code and the HIP runtime are not
# instrumented): build/tsan/libray_mi355x.so + build/tsan/ctx_threads, run on a GPU box by tools/gpu.sh tsan
TSAN := -fsanitize=thread -g
build/tsan/ctx_threads: tools/ctx_threads.cpp $(CSRC)/api.cpp $(CSRC)/multi_gpu.cpp $(CSRC)/host_build.cpp $(OBJ)/render_kernels.o $(OBJ)/bvh_build.o
	@mkdir -p build/tsan
	$(HIPCC) $(HIPFLAGS) $(TSAN) -c $(CSRC)/api.cpp -o build/tsan/api.o
	$(HIPCC) $(HIPFLAGS) $(TSAN) -c $(CSRC)/multi_gpu.cpp -o build/tsan/multi_gpu.o
	/opt/rocm/lib/llvm/bin/clang++ $(HOSTFLAGS) $(TSAN) -c $(CSRC)/host_build.cpp -o build/tsan/host_build.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(TSAN) -o build/tsan/libray_mi355x.so $(OBJ)/render_kernels.o $(OBJ)/bvh_build.o build/tsan/api.o build/tsan/multi_gpu.o build/tsan/host_build.o -ldl
	/opt/rocm/lib/llvm/bin/clang++ -O1 -std=c++17 $(TSAN) -Iinclude -pthread -o $@ tools/ctx_threads.cpp -Lbuild/tsan -lray_mi355x -Wl,-rpath,'$$ORIGIN'

# ... and once more with the locks compiled out (-DRT_NO_CONTEXT_LOCK): what the sanitizer says about the library WITHOUT them (the control run)
build/tsan_nolock/ctx_threads: tools/ctx_threads.cpp $(CSRC)/api.cpp $(CSRC)/multi_gpu.cpp $(CSRC)/host_build.cpp $(OBJ)/render_kernels.o $(OBJ)/bvh_build.o
	@mkdir -p build/tsan_nolock
	$(HIPCC) $(HIPFLAGS) $(TSAN) -DRT_NO_CONTEXT_LOCK -c $(CSRC)/api.cpp -o build/tsan_nolock/api.o
	$(HIPCC) $(HIPFLAGS) $(TSAN) -DRT_NO_CONTEXT_LOCK -c $(CSRC)/multi_gpu.cpp -o build/tsan_nolock/multi_gpu.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(TSAN) -o build/tsan_nolock/libray_mi355x.so $(OBJ)/render_kernels.o $(OBJ)/bvh_build.o build/tsan_nolock/api.o build/tsan_nolock/multi_gpu.o build/tsan/host_build.o -ldl
	/opt/rocm/lib/llvm/bin/clang++ -O1 -std=c++17 $(TSAN) -Iinclude -pthread -o $@ tools/ctx_threads.cpp -Lbuild/tsan_nolock -lray_mi355x -Wl,-rpath,'$$ORIGIN'

build/first_call_probe: tools/first_call_probe.c include/ray.h $(LIB)
	@mkdir -p build
	$(CC) -O2 -std=gnu99 -Wall -Iinclude -o $@ tools/first_call_probe.c -Lraytracers_amd -lray_mi355x -Wl,-rpath,'$$ORIGIN/../raytracers_amd'

tools: build/first_call_probe build/ctx_threads build/rtbench build/issue_peak build/queue_check build/donate_check build/treelet_probe build/hip_touch build/cull_bound_check build/cull_pooled

oracle:
	$(MAKE) -s -C oracle

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all tools oracle clean
