#!/bin/bash
# tools/lab/attn_v2/build.sh -> align_anything_amd/libaa_hip_v2.so : the regular library with attention.o replaced by attention_v2.hip (which includes attention.hip)
set -e
cd "$(dirname "$0")/../../.."
C=align_anything_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -I$C -Iinclude "$@" -c tools/lab/attn_v2/attention_v2.hip -o /tmp/attention_v2.o
objs=$(ls $C/build/*.o | grep -v '/attention.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/attention_v2.o -ldl -o align_anything_amd/libaa_hip_v2.so
echo built align_anything_amd/libaa_hip_v2.so
