// The 16x16x32 forms of the direct-A conv (conv1d_f16x3_da_kernel<.., S16 = true>: 11-tap snake convs, 192- and 128-column tiles),
// compiled beside the others.
#define KX_DA_S16 1
#include "conv_f16x3_da.hip"
