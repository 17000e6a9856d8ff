// The 2 x 2-wave forms of the direct-A conv's unrolled main loop (conv1d_f16x3_da_kernel<.., W2 = true>), compiled beside the others.
#define KX_DA_W2 1
#include "conv_f16x3_da.hip"
