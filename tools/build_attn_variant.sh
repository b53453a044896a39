#!/bin/bash
# lab builds of the attention kernels: tools/build_attn_variant.sh <tag> [-DFLAG ...] -> align_anything_amd/libaa_hip_<tag>.so
# (every other object is taken from the last regular build; select the library with AA_HIP_LIB, tools/attn_lab.py does)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
C=align_anything_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result "$@" -c $C/attention.hip -o /tmp/attention_$tag.o
objs=$(ls $C/build/*.o | grep -v '/attention.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/attention_$tag.o -ldl -o align_anything_amd/libaa_hip_$tag.so
echo built align_anything_amd/libaa_hip_$tag.so
