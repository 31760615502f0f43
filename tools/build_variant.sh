#!/bin/bash
# One-off variant of the FCN translation unit for A/B runs through IVFRONT_LIB: tools/build_variant.sh <name> [-DMACRO=value ...]
# compiles iv_slam_amd/csrc/ivf_fcn.hip with the extra flags and links it with the product's other objects into var/<name>.so
# (var/ is not tracked; it travels to the GPU box with the snapshot).  `make -C iv_slam_amd/csrc` first.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
C=$R/iv_slam_amd/csrc; B=$C/build_var/$name
mkdir -p $B $R/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-variable "$@" -c -o $B/ivf_fcn.o $C/ivf_fcn.hip
printf 'extern "C" const char* ivf_build_id(void) { return "variant:%s"; }\nextern "C" const char* ivf_build_flags(void) { return "%s"; }\n' "$name" "$*" > $B/build_id.cpp
g++ -O1 -fPIC -c -o $B/build_id.o $B/build_id.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/var/$name.so $C/build/ivf_kernels.o $C/build/ivf_api.o $B/ivf_fcn.o $C/build/ivf_rectify.o $C/build/ivf_track.o $B/build_id.o -Wl,-rpath,/opt/rocm/lib
echo "built var/$name.so ($*)"
