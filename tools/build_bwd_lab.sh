#!/bin/bash
# timing-only lab builds of the attention backward (csrc/attention.hip AA_BWD_LAB / AA_BWD_LAB_ONLY): one library per variant, for tools/attn_lab.py
set -e
cd "$(dirname "$0")/.."
v() { bash tools/build_attn_variant.sh "$@" > /dev/null && echo "  built $1"; }
v lab_base &
v lab_dq    -DAA_BWD_LAB_ONLY=1 &
v lab_dkv   -DAA_BWD_LAB_ONLY=2 &
v lab_dq_nosm  -DAA_BWD_LAB_ONLY=1 -DAA_BWD_LAB=1 &
wait
v lab_dq_nodma -DAA_BWD_LAB_ONLY=1 -DAA_BWD_LAB=2 &
v lab_dq_nop1  -DAA_BWD_LAB_ONLY=1 -DAA_BWD_LAB=4 &
v lab_dq_nop2  -DAA_BWD_LAB_ONLY=1 -DAA_BWD_LAB=8 &
v lab_dq_nobar -DAA_BWD_LAB_ONLY=1 -DAA_BWD_LAB=18 &
wait
v lab_dkv_nosm  -DAA_BWD_LAB_ONLY=2 -DAA_BWD_LAB=1 &
v lab_dkv_nodma -DAA_BWD_LAB_ONLY=2 -DAA_BWD_LAB=2 &
v lab_dkv_nop1  -DAA_BWD_LAB_ONLY=2 -DAA_BWD_LAB=4 &
v lab_dkv_nop2  -DAA_BWD_LAB_ONLY=2 -DAA_BWD_LAB=8 &
wait
v lab_dkv_nobar -DAA_BWD_LAB_ONLY=2 -DAA_BWD_LAB=18 &
v lab_dkv_nomfma -DAA_BWD_LAB_ONLY=2 -DAA_BWD_LAB=12 &
v lab_dkv_onlymfma -DAA_BWD_LAB_ONLY=2 -DAA_BWD_LAB=19 &
v lab_dq_onlymfma -DAA_BWD_LAB_ONLY=1 -DAA_BWD_LAB=19 &
wait
ls align_anything_amd/libaa_hip_lab_*.so | wc -l
