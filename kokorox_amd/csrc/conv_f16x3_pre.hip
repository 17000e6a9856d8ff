// Pre-split input images for the direct-A conv kernels (conv_f16x3_da.hip, "PRE").
//
// Why (profiles/r05_pmc_sq_conv_classes.txt, r04_pmc_conv_traffic_f16x3.json): a direct-A workgroup transforms the window it
// stages -- AdaIN affine, activation, f16 hi / lo split, 14 - 27 vector instructions per element -- and a layer with R row tiles
// of 128 output channels does that R times per window: 8 times in the 1024-row decoder convs, 20 / 6 times in the polyphase
// upsamplers, whose run-time-tap form cannot even hide it between its MFMAs.  For those layers the transform is done ONCE here,
// by an elementwise pass that writes the image in the very layout the kernels keep in LDS; staging a chunk is then five or six
// 16-byte loads and LDS writes per lane.  The pass costs one read and one write of the tensor (HBM-bound, these tensors are
// F or 2F or 20F columns long); the rule below takes the layers where that is cheaper than the repeated transform.
//
// Same arithmetic, operation for operation, as emit8 / xform_* of conv_f16x3_da.hip (fma of (x - mean) with scale and shift,
// activation, one multiply by the pre-scale, split_pair): a layer's bits do not depend on which path staged it (tested:
// tests/test_gpu_kernels.py::test_pre_split_image_gives_the_same_bits).
#include "conv_f16x3_common.h"

namespace kx {

static int pre_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Which layers get a pre-split image: by the layer's shape alone (never by the batch, so an utterance's bits do not depend on
// what it is batched with -- and the arithmetic is the same either way).  KX_PRE_ROWS: smallest row count (0 = never).
bool conv16_pre_shape(int BM, int rows, int K, int dil, int stride, int act, int in_up2) {
    // >= 4 row tiles per window.  At batch 64 the 512-row layers (the predictor's F0 / N convs, decode.3) gain from it what their six
    // extra passes cost (profiles/r05_experiments_not_kept.txt: 113.32 vs 113.28 ms); at batch 1 they then take the narrow form
    // (conv_f16x3_dapn.hip: 64 workgroups instead of 16 on the forward's critical path), which is what the rule is set by.
    static const int min_rows = pre_env_int("KX_PRE_ROWS", 512);
    return min_rows > 0 && BM == 128 && rows >= min_rows && stride == 1 && K >= 2 && !in_up2 && act != ACT_SNAKE &&
           conv16_da_eligible(BM, K, dil, stride, 0) && conv16_use_da(BM, K, dil, stride, 0);
}

size_t conv16_pre_image_bytes(int Cin, int x_ld) { return (size_t)((Cin + CK16 - 1) / CK16) * 64 * (size_t)x_ld; }

// one thread = one column x one channel octet of one chunk: 8 coalesced 4-byte reads (a row segment per channel), two 16-byte
// writes (hi and lo plane).  Columns past the utterance's length are not written (the conv kernels mask them by position).
template <int ACT>
__global__ __launch_bounds__(256) void split_image_kernel(const float* __restrict__ x, long x_bs, int x_ld, int C, LenMap len,
                                                          const float* __restrict__ nmean, const float* __restrict__ nscale,
                                                          const float* __restrict__ nshift, int n_bs, float slope, float prescale,
                                                          uint4* __restrict__ img, long img_bs_u4, int img_ld) {
    const int b = blockIdx.z, g = blockIdx.y & 1, ch = blockIdx.y >> 1;
    const int L = len.lens[b] * len.mul + len.add;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= L) return;
    const float* xb = x + (long)b * x_bs;
    const int cmax = C - 1;
    unsigned hp[4], lp[4];
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
        float y2[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = ch * CK16 + g * 8 + 2 * c2 + q;
            const int cc = c < cmax ? c : cmax;
            const bool cvalid = c <= cmax;
            // (the parameter forms of load_params: no norm = (x - 0) * 1 + 0; channels past the end = scale and shift 0)
            const float m = nmean ? nmean[(long)b * n_bs + cc] : 0.f;
            const float sc = cvalid ? (nscale ? nscale[(long)b * n_bs + cc] : 1.f) : 0.f;
            const float sh = (cvalid && nshift) ? nshift[(long)b * n_bs + cc] : 0.f;
            const float v = xb[(long)cc * x_ld + p];
            const float y = in_act<ACT>(__builtin_fmaf(v - m, sc, sh), slope, 1.f, 1.f);
            y2[q] = y * prescale;
        }
        split_pair(y2[0], y2[1], hp[c2], lp[c2]);
    }
    uint4* ib = img + (long)b * img_bs_u4 + (long)ch * 4 * img_ld + p;
    ib[(long)(0 * 2 + g) * img_ld] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
    ib[(long)(1 * 2 + g) * img_ld] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
}

void launch_split_image(const ConvArgs& a, int B, int Lmax, void* img, long img_bs, hipStream_t s) {
    KX_REQUIRE(img != nullptr && a.x != nullptr && B > 0 && B < 65536 && Lmax > 0, "split image: bad argument");
    KX_REQUIRE(a.act == ACT_NONE || a.act == ACT_LEAKY, "split image: identity or leaky activations only");
    KX_REQUIRE(!a.in_up2 && img_bs % 16 == 0 && (size_t)img_bs >= conv16_pre_image_bytes(a.Cin, a.x_ld), "split image: bad image geometry");
    const int n_chunks = (a.Cin + CK16 - 1) / CK16;
    dim3 grid((Lmax + 255) / 256, 2 * n_chunks, B);
    KX_REQUIRE(grid.y < 65536, "split image: too many channels");
    uint4* im = static_cast<uint4*>(img);
    if (a.act == ACT_LEAKY)
        hipLaunchKernelGGL(split_image_kernel<ACT_LEAKY>, grid, dim3(256), 0, s, a.x, a.x_bs, a.x_ld, a.Cin, a.in_len, a.nmean, a.nscale,
                           a.nshift, a.n_bs, a.slope, a.x_prescale, im, img_bs / 16, a.x_ld);
    else
        hipLaunchKernelGGL(split_image_kernel<ACT_NONE>, grid, dim3(256), 0, s, a.x, a.x_bs, a.x_ld, a.Cin, a.in_len, a.nmean, a.nscale,
                           a.nshift, a.n_bs, a.slope, a.x_prescale, im, img_bs / 16, a.x_ld);
    KX_HIP(hipGetLastError());
}

}  // namespace kx
