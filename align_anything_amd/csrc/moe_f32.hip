// fp32 instantiation of the mixture-of-experts routing / token-movement kernels (parity mode): same source as moe.hip with
// elem_t = float; every entry point gets an _f32 twin (include/aa_hip_f32.h).
#define AA_ELEM_F32 1
#include "moe.hip"
