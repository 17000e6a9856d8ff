// The f16f8 forms of the direct-A conv's 16x16x32 loop (conv1d_f16x3_da_kernel<.., S16 = true, .., F8 = true>: a_hi b_hi on the f16
// MFMA, the two cross terms of a product on v_mfma_scale_f32_16x16x128_f8f6f4), compiled beside the others.  Opt-in (KOKOROX_CONV=f16f8).
#define KX_DA_S16 1
#define KX_DA_F8 1
#include "conv_f16x3_da.hip"
