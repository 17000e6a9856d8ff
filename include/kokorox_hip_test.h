/* Test hooks of libkokorox_hip (tests/ only): stand-alone runs of single kernels on host arrays and two process-wide test
 * switches.  They live in a library of their own, libkokorox_hip_test.so (kokorox_amd/csrc/test_hooks.hip), which links against
 * libkokorox_hip.so; the production library exports none of them. */
#ifndef KOKOROX_HIP_TEST_H
#define KOKOROX_HIP_TEST_H
#include "kokorox_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Stand-alone run of the conv1d MFMA kernel on host arrays (x [B,Cin,L], w [Cout,Cin,k]
 * or, for transposed = 1, [Cin,Cout,k]); y must hold B*Cout*Lout floats.  act: 0 none,
 * 1 leaky(slope), 2 snake(alpha[Cin]); norm = optional [3,B,Cin] (mean, scale, shift);
 * mode as in kx_set_conv_mode, plus 2 = f16x3 on the LDS-DMA kernel forms only (conv_f16x3.hip: what the direct-A
 * kernels are compared with bit for bit) and 3 = f16x3 through the direct-A kernel whatever the grid; + 0x100 (modes 1 and 3,
 * here and in the other conv hooks) = the input staged through a pre-split image (conv_f16x3_pre.hip) whatever the row count. */
int kx_test_conv1d(int device_id, const float* x, int B, int Cin, int L, const float* w,
                   const float* bias, int Cout, int k, int stride, int pad, int dil,
                   int transposed, int act, float slope, const float* alpha,
                   const float* norm, float* y, int Lout, int mode, char* err, size_t err_len);

/* A polyphase transposed conv (k = 2 stride, pad = stride / 2) as the generator's upsamplers run it: x [B,Cin,L], w [Cin,Cout,k],
 * fused leaky input activation, residual [B,Cout,Ly] added in the scatter store; up_off = 1: the output starts at column 1 of y
 * and column 0 is its reflection (ReflectionPad1d((1, 0)), the last upsampler), Ly = Lout + up_off, Lout = (L-1) stride - 2 pad + k.
 * mode as kx_test_conv1d, + 0x100 = through a pre-split input image. */
int kx_test_conv_transpose(int device_id, const float* x, int B, int Cin, int L, const float* w, const float* bias, int Cout,
                           int stride, int act, float slope, const float* resid, int up_off, float* y, int mode, char* err,
                           size_t err_len);

/* The epilogue forms of the same kernel on a stride-1 conv: y = (conv + bias + resid [+ y]) * out_mul / out_div
 * with y [B,Cout,Lout] (Lout = L + 2 pad - dil (k-1)) read as the running sum when accumulate != 0, and - when
 * stats_out [B,Cout,2] is given - the fused InstanceNorm partial sums (sum, sum of squares of each stored row). */
int kx_test_conv1d_epilogue(int device_id, const float* x, int B, int Cin, int L, const float* w,
                            const float* bias, int Cout, int k, int pad, int dil, const float* resid,
                            int accumulate, float out_mul, float out_div, float* y, float* stats_out,
                            int mode, char* err, size_t err_len);

/* Both at once, as the generator's resblock convs run: fused AdaIN affine + activation on the input AND the epilogue forms
 * (residual / running sum / scale / fused statistics), on a RAGGED batch when lens [B] is given (utterance b is lens[b]
 * columns long: columns past its output length stay as they were, its statistics cover its own columns only) and, with
 * pad_ld & 1, on rows padded to a multiple of 32 floats as the model lays them out (input and residual padding is NaN,
 * the output padding holds a sentinel that must survive); pad_ld & 2 = the flat list of live tiles the model hands the
 * direct-A kernels on a batch of more than one utterance instead of a (longest length) x B grid.  Stride 1, not transposed. */
int kx_test_conv1d_full(int device_id, const float* x, int B, int Cin, int L, const int32_t* lens, int pad_ld,
                        const float* w, const float* bias, int Cout, int k, int pad, int dil, int act, float slope,
                        const float* alpha, const float* norm, const float* resid, int accumulate, float out_mul,
                        float out_div, float* y, float* stats_out, int mode, char* err, size_t err_len);

/* Stand-alone bidirectional LSTM (hidden 256): x [B,L,n_in] -> y [B,L,512]. */
int kx_test_lstm(int device_id, const float* x, int B, int L, int n_in, const float* w_ih,
                 const float* w_hh, const float* b_ih, const float* b_hh, const float* w_ih_r,
                 const float* w_hh_r, const float* b_ih_r, const float* b_hh_r, float* y,
                 char* err, size_t err_len);

/* Stand-alone ALBERT self-attention (12 heads x 64): qkv [B][2304][T] (rows Q | K | V, head-major, time
 * contiguous) with per-utterance valid lengths -> ctx [B][768][T]; columns >= lens[b] are left untouched. */
int kx_test_attention(int device_id, const float* qkv, const int32_t* lens, int B, int T, float* ctx,
                      char* err, size_t err_len);

/* Stand-alone harmonic source: f0 [B, 2F] -> har_source [B, 600F] (bit-exact phase). */
int kx_test_source(int device_id, const float* f0, int B, int F2, const float* lin_w,
                   float lin_b, uint64_t seed, uint64_t utt_base, int noise_off, float* out,
                   char* err, size_t err_len);

/* Fault injection for the two-CU LSTM recurrence (process-wide, test only): nth > 0 makes the nth following launch of
 * the pair kernel, and every later one, lose the second half of each pair and poll with a short limit, so the call it
 * belongs to must fail with KX_ERR_DEVICE (the bounded wait's error path) and the model must fall back to the one-CU
 * kernel; 0 switches it off.  nth = 6 hits the frame-axis LSTM of a forward, after the mid-way error check.
 * Arms only in a process whose environment has KX_TEST_HOOKS=1 (KX_ERR_STATE otherwise). */
int kx_test_lstm_fault(int nth);

/* Workgroups per (utterance, direction) of the LSTM recurrence (process-wide, test only): 2 or 4 = that resident-weights form
 * whatever the batch, 1 = the streaming fall-back (one workgroup), 0 = by batch size (four while 8 B workgroups fit the CUs).
 * All three give the same bits. */
int kx_test_lstm_parts(int n);

#ifdef __cplusplus
}
#endif
#endif /* KOKOROX_HIP_TEST_H */
