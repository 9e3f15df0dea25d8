# Builds every native artefact in-tree:
#   ray-tracer_amd/librtnative.so  the gfx950 HIP library behind the C-ABI (include/rt_native.h)
#   ray-tracer_amd/librthost.so    the C++ host side above the C-ABI (scene builder + frame driver)
#   oracle/librtoracle.so          the CPU oracle (test infrastructure only)
# hipcc cross-compiles for gfx950 without a GPU.
HIPCC ?= hipcc
CXX ?= g++
PKG := ray-tracer_amd
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function
CXXFLAGS := -O2 -std=c++17 -fPIC -fopenmp -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unused-parameter

all: native host oracle

native: $(PKG)/librtnative.so
host: $(PKG)/librthost.so
oracle:
	$(MAKE) -C oracle

$(PKG)/librtnative.so: $(PKG)/csrc/rt_native.hip $(PKG)/csrc/rt_kernels.hip $(PKG)/csrc/rt_kernels_ris.hip $(PKG)/csrc/rt_device.h include/rt_native.h include/rt_types.h include/rt_detmath.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(PKG)/csrc/rt_native.hip

$(PKG)/librthost.so: $(PKG)/host/scene_builder.cpp $(PKG)/host/frame_driver.cpp $(PKG)/host/ris_driver.cpp $(PKG)/host/scene_cache.cpp $(PKG)/host/scene_internal.h $(PKG)/host/rth_api.h $(PKG)/host/unity_math.h $(PKG)/host/unity_shim.h include/rt_native.h include/rt_types.h $(PKG)/librtnative.so
	$(CXX) $(CXXFLAGS) -shared -o $@ $(PKG)/host/scene_builder.cpp $(PKG)/host/frame_driver.cpp $(PKG)/host/ris_driver.cpp $(PKG)/host/scene_cache.cpp -L$(PKG) -lrtnative -ldl -Wl,-rpath,'$$ORIGIN'

clean:
	rm -f $(PKG)/librtnative.so $(PKG)/librthost.so
	$(MAKE) -C oracle clean

.PHONY: all native host oracle clean
