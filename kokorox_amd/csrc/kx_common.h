// Internal declarations shared by the kokorox_hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <stdexcept>
#include <string>

#include <atomic>
#include <mutex>

#include "kx_error.h"

namespace kx {

constexpr int KX_MAX_DEVICES = 64;

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel, and a process may hold one model per GPU,
// each driven from a host thread of its own (kx_create_replicas, the dispatcher's workers): one of these per launcher
// (function-local static), raised per device only as far as a launch needs, race-free.
struct DynLdsLimit {
    std::atomic<size_t> limit[KX_MAX_DEVICES];
    std::mutex mu;
    DynLdsLimit() {
        for (auto& l : limit) l.store(64 * 1024);  // what a kernel may use without the attribute
    }
    void ensure(const void* kern, size_t lds) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= KX_MAX_DEVICES) throw Error(3, "dynamic LDS limit: bad current device");
        if (lds <= limit[dev].load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> lk(mu);
        if (lds <= limit[dev].load(std::memory_order_relaxed)) return;
        const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) throw Error(3, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
        limit[dev].store(lds, std::memory_order_release);
    }
};

#define KX_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            throw kx::Error(3, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" +  \
                                   __FILE__ + ":" + std::to_string(__LINE__) + ")");       \
    } while (0)

#define KX_REQUIRE(cond, msg)                                 \
    do {                                                      \
        if (!(cond)) throw kx::Error(1, std::string(msg));    \
    } while (0)

// Valid length of utterance b for a tensor = lens[b] * mul + add (lens = tokens or frames).
struct LenMap {
    const int* lens;
    int mul, add;
};

// ---- conv1d as implicit GEMM on the f32 MFMA (conv_mfma.hip) -------------------------
constexpr int CONV_CK = 8;  // input channels per K-chunk

enum ConvAct { ACT_NONE = 0, ACT_LEAKY = 1, ACT_SNAKE = 2 };
enum ConvStore { ST_NORMAL = 0, ST_TMAJOR = 1, ST_UPSCATTER = 2 };
enum ConvEpi { EPI_NONE = 0, EPI_GELU_NEW = 1 };

struct ConvArgs {
    // input [B][Cin][x_ld]
    const float* x;
    long x_bs;
    int x_ld;
    int Cin;
    LenMap in_len;   // valid (virtual, after in_up2) input length
    LenMap out_len;  // valid output length (ST_UPSCATTER: un-shifted upsampled length)
    // packed weights [co_tile][chunk][k][8][BM] and bias [Cout]
    const float* w;
    const float* bias;
    // optional per (b, ci) affine (x - mean) * scale + shift, each [B][n_bs]
    const float* nmean;
    const float* nscale;
    const float* nshift;
    int n_bs;
    int act;
    float slope;
    const float* alpha;  // snake, [Cin]
    int K, dil, stride, pad;
    int in_up2;  // read x[p >> 1] (nearest x2 upsample of the input)
    int Cout;    // GEMM rows (ST_UPSCATTER: up_s * up_cout)
    int n_chunks;
    // output [B][Cout][y_ld] (ST_TMAJOR: [B][L][y_ld])
    float* y;
    long y_bs;
    int y_ld;
    const float* resid;
    long r_bs;
    int r_ld;
    int accum;
    float out_mul;
    float out_div;
    int epi;
    int store;
    int up_s, up_pad, up_off, up_reflect, up_cout;
    // f16x3 path (conv_f16x3.hip): split-f16 weight image, 16-channel chunks, 2^-ws to undo the weight scale
    const void* w16;
    const void* w16b;  // the bf16 form of the same image (prec1 == 2 launches of the direct-A kernel swap it in), or null
    const void* w8x;   // f16f8 mode (CONV_F16F8): the 8-bit image of the weights' cross-term operands (launch_pack_conv8x), or null
    int epi_stream;    // the direct-A kernels' interior stores and residual loads non-temporal (the output tensor is far beyond L2 + MALL)
    int n_chunks16;
    float w_unscale;
    float x_prescale;  // f16x3: power of two applied to the transformed input before the hi/lo split (w_unscale carries
                       // its inverse); 1 unless a layer's activations are so small that their low halves would be lost
    // k = 1 GEMMs over a short axis: columns of all B utterances form one merged space of B * merge_T
    // columns (utterance = col / merge_T); 0 = off.  Requires K = 1, stride 1, pad 0, no norm, no in_up2.
    int merge_T;
    int merge_B;
    float2* stat_part;  // optional [B][Cout][stat_tiles] partial (sum, sum of squares) of the stored values
    int stat_tiles;
    int xcd_swizzle;               // one-role f16x3 kernel: contiguous column tiles per XCD (1 unless KX_XCD_SWIZZLE=0)
    int ws_force;                  // test hook: 1 = the LDS-DMA kernel forms only (no direct-A kernel), 2 = the direct-A kernel whatever the grid
    unsigned long long* stamps;  // diagnostic build only: per-workgroup {t0,t1,t2,t3,hw_id,xcc_id,0,0}
    int prec1;  // direct-A conv: the reduced-precision forms, opt-in: 1 = one f16 MFMA per product (KOKOROX_CONV=f16), 2 = one bf16
                // MFMA per product on a bf16 weight image (KOKOROX_CONV=bf16)
    int dephase_cycles, dephase_mode;  // direct-A conv: start delay of half of the first round of workgroups (0 = off)
    int dbg;  // timing ablations (env KX_DBG): 1 skip input staging, 2 skip weight copies, 4 skip MFMA, 8 skip epilogue
    // Flat tile list of a ragged batch (direct-A kernels): tile_prefix[b] = column tiles of the utterances before b (so
    // tile_prefix[B] = all of them); the grid is then ONE dimension of tile_prefix[B] * flat_ny workgroups, none of them
    // dead, and the XCD-aware order runs over the whole list.  null = the (max_cols / BN) x row tiles x B grid with
    // early-exit workgroups.  (The dense grid deals each XCD the same eighth of EVERY utterance's slab, so on a ragged
    // batch the XCDs that hold the slabs' tails run empty and the launch takes as long as if every utterance had the
    // longest length: profiles/r04_ragged_attribution.txt.)
    const int* tile_prefix;
    int flat_ny;     // row tiles (the fastest index of the list)
    int flat_B;
    int flat_tiles_host, flat_bn_host;  // host side only: tile_prefix[B] as the host counted it, and the tile width it assumed
    // Pre-split input image (direct-A convs whose window is staged by many row tiles; conv_f16x3_pre.hip): the input after
    // its AdaIN affine, activation, pre-scale and f16 hi / lo split, per utterance [chunk of 16 channels][hi|lo][octet][x16_ld
    // columns][8 halves] = the kernels' LDS image; x16_bs bytes per utterance.  null = the kernel transforms a.x itself.
    const void* x16;
    long x16_bs;
    int x16_ld;
};

// pre-split images (conv_f16x3_pre.hip)
bool conv16_pre_shape(int BM, int rows, int K, int dil, int stride, int act, int in_up2);  // which layers get one: by shape alone
size_t conv16_pre_image_bytes(int Cin, int x_ld);  // per utterance
// writes the image of a.x (a's norm parameters, activation, slope, pre-scale) for B utterances of at most Lmax columns
// the narrow pre-split form for small grids (conv_f16x3_dapn.hip): 32 rows x 128 columns per workgroup, B fragments straight from the image
bool conv16_dapn_eligible(const ConvArgs& a);
void launch_conv1d_f16x3_dapn(const ConvArgs& a, int B, int max_cols, hipStream_t s);
void launch_split_image(const ConvArgs& a, int B, int Lmax, void* img, long img_bs, hipStream_t s);

// tile_prefix of a LenMap for `bn`-column tiles (+ `extra` columns per utterance: the polyphase convs' L + 1), on the device
void launch_tile_prefix(LenMap len, int extra, int bn, int B, int* out, hipStream_t s);
// the launch geometry launch_conv1d_f16x3 will choose for these arguments if it can take a flat tile list: the tile width
// (128 / 192 / 256 columns), or 0 when the launch goes to a kernel without the flat form
int conv16_flat_bn(const ConvArgs& a, int BM, int B, int max_cols);

struct ConvShape {
    int BM;  // 128, 64 or 32
};
inline int conv_pick_bm(int rows) { return rows >= 128 ? 128 : (rows > 32 ? 64 : 32); }
inline int conv_bn(int BM) { return BM == 128 ? 128 : 256; }

void launch_conv1d(const ConvArgs& a, int BM, int B, int max_cols, hipStream_t s);

// weight repack: canonical layouts -> [co_tile][chunk][k][8][BM]
struct PackSrc {
    const float* p[3];  // up to three row-concatenated sources, each [rows_i][Cin][K]
    int rows[3];
};
void launch_pack_conv(const PackSrc& src, float* dst, int Cout, int Cin, int K, int BM, hipStream_t s);
// ConvTranspose1d weight [Cin][Cout][k], k == 2*s  ->  polyphase GEMM rows (p, co), 2 taps
void launch_pack_convT(const float* w, float* dst, int Cin, int Cout, int s, int BM, hipStream_t s_);
size_t packed_conv_floats(int rows, int Cin, int K, int BM);

// f16x3 split path
enum ConvMode { CONV_F32 = 0, CONV_F16X3 = 1, CONV_F16X3_LDS = 2, CONV_F16X3_DA = 3, CONV_F16 = 4, CONV_BF16 = 5, CONV_F16F8 = 6 };  // (2, 3: test hook only: f16x3 kept on the LDS-DMA kernel forms of conv_f16x3.hip / forced through conv_f16x3_da.hip)
void launch_conv1d_f16x3(const ConvArgs& a, int BM, int B, int max_cols, hipStream_t s);
// precision class of a launch, as the shape rules see it: 0 = f16x3, 1 = reduced precision (one MFMA per product), 2 = f16f8 (the
// layer carries an 8-bit cross image)
inline int conv16_pmode(const ConvArgs& a) { return a.prec1 ? 1 : (a.w8x ? 2 : 0); }
// (act / n_chunks16 / pmode: given, the shapes of the direct-A S16 form get its tiles: 192 or 128 columns, 64-column statistics slots)
void conv16_pick_tile(int BM, int max_cols, int B, int Cout, int K, int dil, int stride, int* bn, int* wn,
                      int ws_force = 0, bool stats = false, int act = -1, int n_chunks16 = 0, int pmode = 0);  // (conv_f16x3.hip)
// conv_f16x3_da.hip: the 128 x 256 tile with the weight fragments loaded from global memory straight into registers
bool conv16_use_da(int BM, int K, int dil, int stride, int merged);       // eligible AND switched on (default; KX_DA=0 turns it off)
bool conv16_da_eligible(int BM, int K, int dil, int stride, int merged);  // shape fits the kernel
void launch_conv1d_f16x3_da(const ConvArgs& a, int B, int max_cols, hipStream_t s, int bn = 256);
// conv_f16x3_dag.hip: k = 1 GEMMs (incl. the merged token-axis form) in the direct-A form, 128 x 128 tile
bool conv16_use_dag(const ConvArgs& a, int BM);       // eligible AND switched on (default; KX_DAG=0 turns it off)
bool conv16_dag_eligible(const ConvArgs& a, int BM);
void launch_conv1d_f16x3_dag(const ConvArgs& a, int B, int max_cols, hipStream_t s);
int conv16_cu_count();  // CUs of the current device, or of the launching model's CU partition (conv_f16x3.hip)
int cu_count_override(); // model.hip: the CU partition of the model that is launching on this thread (0 = none)
// shapes that take the 16x16x32 form of the direct-A conv (conv_f16x3_da_s16.hip): 192 / 128-column tiles, 64-column statistics slots
bool conv16_da_s16_shape(int BM, int K, int dil, int stride, int act, int n_chunks16, bool merged, int pmode);
size_t packed_conv16_halves(int rows, int Cin, int K, int BM);
float device_absmax(const float* p, long n, hipStream_t s);
int pick_weight_shift(float absmax);
void launch_pack_conv16(const PackSrc& src, void* dst, int Cout, int Cin, int K, int BM, float wscale, hipStream_t s);
void launch_pack_convT16(const float* w, void* dst, int Cin, int Cout, int sd, int BM, float wscale, hipStream_t s);
// a split-f16 weight image (hi + lo = the f32 weight to 22 bits) -> the same layout with bf16(hi + lo) in the hi slots (CONV_BF16)
void launch_image_to_bf16(const void* w16, void* dst, size_t n_halves, hipStream_t s);
// f16f8 mode: the cross terms a_lo b_hi + a_hi b_lo of the split product on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, twice the
// f16 rate), a_hi b_hi stays on the f16 MFMA.  The weights' side of the cross terms, from a split-f16 image of 128-row tiles:
// [row tile][chunk16][tap group of 4][k-group g = octet + 2 (tap pair)][slot 0 | 1][128 rows][16 B] (a lane's 32-byte operand is
// its row's two slots: taps 4 m + 2 (g >> 1) and + 1; taps >= K are zero; a load instruction reads one slot of 16 consecutive
// rows = 256 contiguous bytes), a slot = four dwords [e4m3(2^7 lo(2d)), e4m3(2^7 lo(2d+1)), e4m3(2^-4 hi(2d)),
// e4m3(2^-4 hi(2d+1))] of the octet's channel pairs d -- byte for byte the partner of the activation image's 8-bit plane
// (split_pair_f8, conv_f16x3_common.h); the products of a slot carry 2^7 (undone by the instruction's block scale).
size_t packed_conv8x_bytes(int rows, int Cin, int K);
void launch_pack_conv8x(const void* w16, void* dst, int rows, int Cin, int K, hipStream_t s);
bool conv16_da_f8_shape(int K, int dil);  // which of the S16 form's shapes have an f16f8 kernel (conv_f16x3_da_f8.hip)
void launch_conv1d_f16x3_da_f8(const ConvArgs& a, int B, int max_cols, hipStream_t s, int bn);

// ---- everything else (kernels_misc.hip) ----------------------------------------------
void launch_vec_add(const float* a, const float* b, float* out, int n, hipStream_t s);
void launch_transpose_whh(const float* whh, float* out, hipStream_t s);  // [1024 rows][256 k] -> [64][1024][4 k]

void launch_albert_embed(const int64_t* ids, long ids_stride, const float* word, const float* type0,
                         const float* pos, float* out, long bs, int ld, const int* lens, int B, int Tmax,
                         int n_vocab, unsigned* bad_id, hipStream_t s);
void launch_embed(const int64_t* ids, long ids_stride, const float* table, int C, float* out, long bs, int ld,
                  const int* lens, int B, int Tmax, int n_vocab, unsigned* bad_id, hipStream_t s);

enum LnMode { LN_PLAIN = 0, LN_AFFINE = 1, LN_ADA = 2 };
// channel layer-norm over C rows of [B][C][ld], in place allowed. ADA: g/be are [B][g_bs], (1+g)*xhat+be
void launch_layernorm_ch(const float* x, float* y, long bs, int ld, int C, LenMap len, int B, int Lmax,
                         float eps, int mode, const float* g, const float* be, int g_bs, float leaky,
                         hipStream_t s);

void launch_attention(const float* qkv, long bs, int ld, float* ctx, long cbs, int cld, const int* lens, int B,
                      int Tmax, hipStream_t s);

struct FcDesc {
    const float* w;  // [n_out][128]
    const float* b;  // [n_out]
    int n_out;
    int style_off;  // 0 or 128 within the 256-float style row
    long out_off;   // into the gamma/beta arena, per utterance
};
void launch_style_fc(const FcDesc* d_desc, int n_desc, const float* styles, float* out, long out_bs, int B,
                     hipStream_t s);

// instance-norm statistics of [B][C][ld] rows -> mean, scale = rstd*(1+gamma), shift = beta
// finalize fused statistics: partials [B][C][tiles] -> mean, scale, shift (same outputs as launch_in_stats)
void launch_stats_finalize(const float2* part, int tiles, int cols_per_tile, int C, LenMap len, int B, const float* gb,
                           long gb_bs, float* mean, float* scale, float* shift, int n_bs, hipStream_t s);
void launch_in_stats(const float* x, long bs, int ld, int C, LenMap len, int B, const float* gb, long gb_bs,
                     float* mean, float* scale, float* shift, int n_bs, float2* raw_out, hipStream_t s);

// diagnostics (real-weights report): absmax and sum of squares of a conv's input after its AdaIN affine
// (out[0] = absmax bits as uint, out[1] = sum of squares, out[2] = element count)
void launch_diag_stats(const float* x, long bs, int ld, int C, LenMap len, int B, int Lmax, const float* nmean,
                       const float* nscale, const float* nshift, int n_bs, float* out3, hipStream_t s);

void launch_style_mix(const float* table, int n_voices, const int* voice_ids, const float* weights, int max_mix,
                      const int* rows, const int* kinds, float* styles, int B, hipStream_t s);
void launch_pack_audio(const float* audio, long audio_ld, const int* frames, int B, int Fmax, int format, void* out,
                       long out_stride_bytes, const long* out_off, hipStream_t s, const int* formats = nullptr);
// Host output buffers of the kx_infer* calls: page-locked and pooled (one asynchronous D2H copy at PCIe rate instead
// of per-utterance pageable copies); host_out_free also accepts plain malloc'd pointers (dispatcher results).
void* host_out_alloc(size_t bytes);
void host_out_free(void* p);
// n pointers into one host_out_alloc'd buffer handed to n owners: each is released with host_out_free, the buffer goes back
// to the pool with the last one (parts[0] may be the buffer's own address)
void host_out_share(void* base, void* const* parts, int n);
// page-locked bytes of shared batch buffers that owners still hold a part of (what the dispatcher caps: a client that keeps one
// result keeps its whole batch's buffer pinned)
size_t host_out_live_bytes();
void launch_fill_style_rows(float* dst, long bs, int ld, int row0, const float* styles, int style_off,
                            const int* lens, int B, int Tmax, hipStream_t s);
void launch_copy_rows(const float* src, long sbs, int sld, float* dst, long dbs, int dld, int rows, LenMap len,
                      int B, int Lmax, hipStream_t s);

// xchg / err_word: exchange buffer (lstm_exchange_bytes(B), any contents) and sticky error word of the two-CU form;
// null = the one-CU streaming kernel
void launch_lstm(const float* gx, long gx_bs, int gx_ld, const float* whhT, float* y, long y_bs, int y_ld,
                 LenMap len, int B, unsigned long long* xchg, unsigned* err_word, hipStream_t s,
                 unsigned* epoch_state = nullptr);  // epoch_state: the buffer's own launch counter (see launch_lstm)
size_t lstm_exchange_bytes(int B);
void lstm_set_parts(int n);       // test hook: 2 / 4 workgroups per (utterance, direction) whatever the batch, 0 = by batch size
void lstm_set_test_fault(int on);  // test hook: the two-CU kernel's partner never shows up (bounded-poll error path)

void launch_duration(const float* logits, long bs, int ld, const float* speeds, int n_speed, const int* lens,
                     const int* pinned, int n_pinned, int* dur, int* frames, int* idx, int idx_ld, int B,
                     hipStream_t s);
void launch_gather_cols(const float* src, long sbs, int sld, float* dst, long dbs, int dld, int C,
                        const int* idx, int idx_ld, const int* frames, int B, int Fmax, hipStream_t s);

// AdainResBlk1d "pool": leaky(norm(x)) -> depth-wise ConvTranspose1d(k3,s2,p1,op1)
void launch_pool_up2(const float* x, long xbs, int xld, int C, const float* mean, const float* scale,
                     const float* shift, int n_bs, float slope, const float* w, const float* bias, float* y,
                     long ybs, int yld, LenMap in_len, int B, int Lmax_in, hipStream_t s);

void launch_source(const float* f0, long f0_bs, const int* frames, int B, int Fmax, const float* lin_w,
                   const float* lin_b, uint64_t seed, uint64_t utt_base, const uint64_t* utt_seeds, int noise_off,
                   float* phase_ws,
                   float* har, long har_bs, hipStream_t s);
enum StftVariant { STFT_ONNX = 0, STFT_TORCH = 1 };  // (kernels_misc.hip: the two STFT / iSTFT pairs)
void launch_stft(const float* har_src, long hs_bs, float* har, long bs, int ld, const int* frames, int B,
                 int Fmax, int variant, hipStream_t s);
void launch_istft_head(const float* cp, long bs, int ld, float* spec_ws, float* audio, long audio_ld,
                       const int* frames, int B, int Fmax, int variant, hipStream_t s);
void init_dft_tables();

}  // namespace kx
