// fp32 instantiation of the block kernels (parity mode): the same source as elementwise.hip compiled with
// elem_t = float, so the bf16 rounding points of the HF graph vanish and every aa_* entry point gets an _f32 twin
// (include/aa_hip_f32.h).  The production path is the bf16 instantiation; this one exists so that the native step
// can be compared against the reference's fp32 CPU trainer at 1e-4 on the loss curve.
#define AA_ELEM_F32 1
#include "elementwise.hip"
