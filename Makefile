# Build of libpcc_nn (gfx950 HIP kernels + C-ABI), the CPU oracle and helper tools.
# hipcc cross-compiles gfx950 without a GPU.  -ffp-contract=off everywhere: the
# distance arithmetic must round like FLANN's L2_Simple (no FMA).
HIPCC      ?= /opt/rocm/bin/hipcc
CC         ?= gcc
CXX        ?= g++
ARCH       ?= gfx950
HIPFLAGS   ?= --offload-arch=$(ARCH) -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -Iinclude -Ipointcloudcomparator_amd/csrc
CSRC       := pointcloudcomparator_amd/csrc
LIBDIR     := pointcloudcomparator_amd/lib
HIP_SRCS   := $(CSRC)/api.hip $(CSRC)/pack.hip $(CSRC)/nn1_brute.hip $(CSRC)/grid.hip $(CSRC)/cellsort.hip $(wildcard $(CSRC)/knn.hip $(CSRC)/cluster.hip $(CSRC)/icp.hip $(CSRC)/voxel.hip $(CSRC)/normals.hip $(CSRC)/region.hip $(CSRC)/sac.hip $(CSRC)/flann_order.hip $(CSRC)/cellsort_mp.hip $(CSRC)/comm.hip $(CSRC)/small.hip)
HDRS       := $(wildcard $(CSRC)/*.hpp) include/pcc_nn.h
HIP_OBJS   := $(patsubst $(CSRC)/%.hip,build/%.o,$(HIP_SRCS))

all: lib oracle hosttest cli prof

lib: $(LIBDIR)/libpcc_nn.so
# the profiling build: same sources with the pair counter compiled in (pcc_index_stats[4]); never the timed library
prof: $(LIBDIR)/libpcc_nn_prof.so
PROF_OBJS  := $(patsubst $(CSRC)/%.hip,build/prof/%.o,$(HIP_SRCS))
build/prof/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/prof
	$(HIPCC) $(HIPFLAGS) -DPCC_COUNT_PAIRS $(EXTRA_HIPFLAGS) -c $< -o $@
$(LIBDIR)/libpcc_nn_prof.so: $(PROF_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(PROF_OBJS) -ldl
oracle: oracle/_build/libpcc_oracle.so
ubench: build/ubench_valu build/ubench_gather build/ubench_scatter
hosttest: build/test_host_mirror build/test_lane_ops build/test_report build/test_libm
cli: build/comparator build/ply_dump build/rgb_segments

build/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(EXTRA_HIPFLAGS) -c $< -o $@

$(LIBDIR)/libpcc_nn.so: $(HIP_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJS) -ldl

oracle/_build/libpcc_oracle.so: oracle/pcc_oracle.c oracle/pcc_oracle.h
	@mkdir -p oracle/_build
	$(CC) -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread -o $@ oracle/pcc_oracle.c -lm

build/ubench_gather: tools/ubench/ubench_gather.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 $< -o $@

build/ubench_scatter: tools/ubench/ubench_scatter.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 $< -o $@

build/ubench_valu: tools/ubench/ubench_valu.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 $< -o $@

build/test_lane_ops: tests/cpp/test_lane_ops.hip $(CSRC)/lane_ops.hpp
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -I$(CSRC) $< -o $@

build/test_host_mirror: tests/cpp/test_host_mirror.cpp include/pcc/point_types.hpp include/pcc/search.hpp include/pcc/comparator_nn.hpp include/pcc/multi_device.hpp include/pcc_nn.h $(LIBDIR)/libpcc_nn.so
	@mkdir -p build
	$(CXX) -std=c++17 -O2 -Wall -pthread -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include $< -o $@ -L$(LIBDIR) -lpcc_nn -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

build/comparator: examples/comparator_main.cpp pointcloudcomparator_amd/host/ply_io.hpp pointcloudcomparator_amd/host/report.hpp include/pcc/multi_device.hpp include/pcc/point_types.hpp include/pcc/search.hpp include/pcc/comparator_nn.hpp include/pcc_nn.h $(LIBDIR)/libpcc_nn.so
	@mkdir -p build
	$(CXX) -std=c++17 -O2 -Wall -pthread -Iinclude -Ipointcloudcomparator_amd/host $< -o $@ -L$(LIBDIR) -lpcc_nn -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

build/test_report: tests/cpp/test_report.cpp pointcloudcomparator_amd/host/report.hpp include/pcc/comparator_nn.hpp include/pcc/search.hpp include/pcc_nn.h $(LIBDIR)/libpcc_nn.so
	@mkdir -p build
	$(CXX) -std=c++17 -O2 -Wall -pthread -Iinclude -Ipointcloudcomparator_amd/host $< -o $@ -L$(LIBDIR) -lpcc_nn -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

build/rgb_segments: tests/cpp/rgb_segments.cpp include/pcc/region_growing_rgb.hpp include/pcc/search.hpp include/pcc/point_types.hpp pointcloudcomparator_amd/host/ply_io.hpp include/pcc_nn.h $(LIBDIR)/libpcc_nn.so
	@mkdir -p build
	$(CXX) -std=c++17 -O2 -Wall -pthread -Iinclude -Ipointcloudcomparator_amd/host $< -o $@ -L$(LIBDIR) -lpcc_nn -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

build/test_libm: tests/cpp/test_libm.cpp $(CSRC)/libm_f32.hpp
	@mkdir -p build
	$(CXX) -std=c++17 -O2 -ffp-contract=off -Wall -I$(CSRC) $< -o $@ -lm

build/ply_dump: tests/cpp/ply_dump.cpp pointcloudcomparator_amd/host/ply_io.hpp include/pcc/point_types.hpp
	@mkdir -p build
	$(CXX) -std=c++17 -O2 -Wall -Iinclude -Ipointcloudcomparator_amd/host $< -o $@

# ---- sanitizers on the host-side code (CPU build only; sanitizers never run on the GPU box) --------------------
# oracle/pcc_oracle.c, csrc/flann_tree.hpp (the PCC_TIES_FLANN tree: build + walk), csrc/rigid_solve.hpp,
# csrc/plane_fit.hpp and host/ply_io.hpp under ASan + UBSan with a CPU-only driver, and the report writer's self-test.
SANFLAGS := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1 -ffp-contract=off
asan: build/asan/asan_driver build/asan/test_flann_tree
	ASAN_OPTIONS=detect_leaks=1 build/asan/asan_driver build/asan
	@echo "asan: clean"

build/asan/pcc_oracle.o: oracle/pcc_oracle.c oracle/pcc_oracle.h
	@mkdir -p build/asan
	$(CC) $(SANFLAGS) -fno-fast-math -pthread -c $< -o $@

build/asan/asan_driver: tests/cpp/asan_driver.cpp build/asan/pcc_oracle.o $(CSRC)/flann_tree.hpp $(CSRC)/rigid_solve.hpp $(CSRC)/plane_fit.hpp pointcloudcomparator_amd/host/ply_io.hpp include/pcc/point_types.hpp
	$(CXX) -std=c++17 $(SANFLAGS) -Wall -pthread -Iinclude -I$(CSRC) -Ipointcloudcomparator_amd/host -Ioracle $< build/asan/pcc_oracle.o -o $@ -lm

build/asan/test_flann_tree: tests/cpp/test_flann_tree.cpp $(CSRC)/flann_tree.hpp
	@mkdir -p build/asan
	$(CXX) -std=c++17 $(SANFLAGS) -Wall -pthread -I$(CSRC) $< -o $@

clean:
	rm -rf build $(LIBDIR)/*.so oracle/_build

.PHONY: all lib prof oracle ubench hosttest cli clean asan
