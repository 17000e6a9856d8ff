// the pre-split-image forms of the direct-A conv kernel (see conv_f16x3_da.hip, "PRE"): a translation unit of their own, compiled beside the others
#define KX_DA_PRE
#include "conv_f16x3_da.hip"
