// The reduced-precision instantiations of the direct-A conv kernel (one f16 MFMA per product: KOKOROX_CONV=f16), compiled
// as a translation unit of their own beside conv_f16x3_da.hip.
#define KX_DA_P1 1
#include "conv_f16x3_da.hip"
