#!/bin/bash
# tools/variant.sh NAME [-DFLAG=VALUE ...] -- build sampler_amd/csrc/variants/NAME.so, a copy of
# libdwx.so compiled with extra flags (kernel A/B experiments).  Use it with DWX_LIB=<path>.
set -e
cd "$(dirname "$0")/../sampler_amd/csrc"
name=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function \
  "$@" -x hip -c -o variants/$name.o dwx_api.cc
# (the graph compiler shares device_types.h with the kernels: same flags)
g++ -O2 -std=c++17 -fPIC -Wall -Wextra -pthread "$@" -c -o variants/$name.gc.o graph_compile.cc
make -s device_build.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so variants/$name.o variants/$name.gc.o device_build.o -pthread
rm -f variants/$name.o variants/$name.gc.o
echo "built sampler_amd/csrc/variants/$name.so"
