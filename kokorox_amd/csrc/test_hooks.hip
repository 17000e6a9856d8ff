// Stand-alone kernel hooks for tests/ (include/kokorox_hip_test.h): libkokorox_hip_test.so, a library of its own that links
// against libkokorox_hip.so -- the production library exports none of them.
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/kokorox_hip.h"
#include "../../include/kokorox_hip_test.h"
#include "model.h"

using kx::Error;
using kx::Model;

#include "kx_handle.h"

#include "api_guard.h"
using kx::guarded;
using kx::guarded_free;
using kx::set_err;

static void check_device(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) throw Error(KX_ERR_DEVICE, "no HIP device is visible (the HIP path has no CPU fallback)");
    if (device_id < 0 || device_id >= n) throw Error(KX_ERR_INVALID, "device id out of range");
    hipDeviceProp_t p;
    KX_HIP(hipGetDeviceProperties(&p, device_id));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        throw Error(KX_ERR_DEVICE, std::string("device is ") + p.gcnArchName + ", this library is built for gfx950 only");
}

namespace {
struct DevMem {
    std::vector<void*> p;
    ~DevMem() {
        for (void* q : p) (void)hipFree(q);
    }
    template <class Tp>
    Tp* get(size_t n) {
        void* q = nullptr;
        KX_HIP(hipMalloc(&q, (n ? n : 1) * sizeof(Tp)));
        p.push_back(q);
        return static_cast<Tp*>(q);
    }
    template <class Tp>
    Tp* up(const Tp* h, size_t n) {
        Tp* d = get<Tp>(n);
        KX_HIP(hipMemcpy(d, h, n * sizeof(Tp), hipMemcpyHostToDevice));
        return d;
    }
};
}  // namespace

extern "C" {

int kx_test_lstm_fault(int nth) {
    // fault injection is a process-wide switch: it arms only when the environment says this is a test process
    const char* e = getenv("KX_TEST_HOOKS");
    if (!e || strcmp(e, "1") != 0) return KX_ERR_STATE;
    kx::lstm_set_test_fault(nth);
    return KX_OK;
}

int kx_test_lstm_parts(int n) {
    kx::lstm_set_parts(n);
    return KX_OK;
}

// ---- stand-alone kernel hooks for tests/ ------------------------------------------------------

namespace {
struct ConvTestExtra {  // epilogue forms beyond bias: residual, accumulate into y, scale / divide, fused statistics
    const float* resid = nullptr;
    int accum = 0;
    float out_mul = 1.f, out_div = 1.f;
    float* stats_out = nullptr;  // [B][Cout][2] = sum, sum of squares over the stored row
    const int32_t* lens = nullptr;  // [B] valid input columns per utterance (ragged batch); null = all L
    int pad_ld = 0;              // rows padded to a multiple of 32 floats as in the model (x padding = NaN, y padding checked)
    int flat = 0;                // ragged batches: the flat tile list the model gives the direct-A kernels (ConvArgs::tile_prefix)
    int up_off = 0, up_reflect = 0;  // transposed: the output starts at column up_off of y (and column 0 = reflection of column 1)
};
}  // namespace

static int test_conv1d_impl(int device_id, const float* x, int B, int Cin, int L, const float* w, const float* bias, int Cout,
                            int k, int stride, int pad, int dil, int transposed, int act, float slope, const float* alpha,
                            const float* norm, float* y, int Lout, int mode, const ConvTestExtra& ex, char* err,
                            size_t err_len) {
    return guarded_free(err, err_len, [&] {
        check_device(device_id);
        KX_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0 && L > 0 && Lout > 0 && k > 0, "test_conv1d: bad argument");
        KX_HIP(hipSetDevice(device_id));
        DevMem dm;
        std::vector<int> lens(B, 1);
        // bit 8 of the mode: stage the input through a pre-split image (conv_f16x3_pre.hip), whatever the layer's row count
        const bool pre = (mode & 0x100) != 0;
        // bit 9: f16f8 -- the layer gets the 8-bit cross image of its weights (the S16 form's shapes then run the f16f8 kernel)
        const bool f8 = (mode & 0x200) != 0;
        mode &= 0xff;
        KX_REQUIRE(mode >= kx::CONV_F32 && mode <= kx::CONV_F16X3_DA, "test_conv1d: mode must be 0, 1, 2 or 3");
        KX_REQUIRE(!pre || mode == kx::CONV_F16X3 || mode == kx::CONV_F16X3_DA, "test_conv1d: pre-split images exist for the direct-A kernels only");
        kx::ConvArgs a{};
        a.ws_force = mode == kx::CONV_F16X3_LDS ? 1 : (mode == kx::CONV_F16X3_DA ? 2 : 0);
        // row strides: the caller's dense rows, or (pad_ld) the model's: a multiple of 32 floats, input padding poisoned
        const int Ly = Lout + ex.up_off;  // columns of y (Lout: the conv's own output length)
        KX_REQUIRE(ex.up_off == 0 || (transposed && ex.up_off == 1 && !ex.pad_ld), "test_conv1d: an output offset comes with transposed convs");
        const int x_ld = ex.pad_ld ? (L + 31) & ~31 : L, y_ld = ex.pad_ld ? (Lout + 31) & ~31 : Ly;
        const float poison = std::nanf(""), sentinel = -12345.5f;
        auto padded = [&](const float* src, int rows_total, int len, int ld, float fill) {
            std::vector<float> v((size_t)rows_total * ld, fill);
            for (int r = 0; r < rows_total; ++r) std::memcpy(&v[(size_t)r * ld], src + (size_t)r * len, (size_t)len * 4);
            return v;
        };
        if (ex.pad_ld) {
            const std::vector<float> xp = padded(x, B * Cin, L, x_ld, poison);
            a.x = dm.up(xp.data(), xp.size());
        } else
            a.x = dm.up(x, (size_t)B * Cin * L);
        a.x_bs = (long)Cin * x_ld;
        a.x_ld = x_ld;
        a.Cin = Cin;
        if (ex.lens) {  // ragged batch: utterance b is lens[b] columns long (stride-1 convs: the output shrinks by L - Lout)
            KX_REQUIRE(!transposed && stride == 1, "test_conv1d: ragged lengths with stride-1 convs only");
            for (int b = 0; b < B; ++b) {
                KX_REQUIRE(ex.lens[b] >= 1 && ex.lens[b] <= L && ex.lens[b] + (Lout - L) >= 1, "test_conv1d: lens out of range");
                lens[b] = ex.lens[b];
            }
        }
        int* d_one = dm.up(lens.data(), B);
        a.in_len = ex.lens ? kx::LenMap{d_one, 1, 0} : kx::LenMap{d_one, 0, L};
        a.out_len = ex.lens ? kx::LenMap{d_one, 1, Lout - L} : kx::LenMap{d_one, 0, Lout};
        a.n_chunks = (Cin + kx::CONV_CK - 1) / kx::CONV_CK;
        const float* dw = dm.up(w, (size_t)Cout * Cin * k);
        int BM, rows;
        float* packed;
        if (!transposed) {
            rows = Cout;
            BM = kx::conv_pick_bm(rows);
            packed = dm.get<float>(kx::packed_conv_floats(rows, Cin, k, BM));
            kx::PackSrc src{{dw, nullptr, nullptr}, {Cout, 0, 0}};
            kx::launch_pack_conv(src, packed, Cout, Cin, k, BM, nullptr);
            a.K = k;
            a.stride = stride;
            a.pad = pad;
            a.dil = dil;
            a.store = kx::ST_NORMAL;
            a.up_cout = 1;
        } else {
            KX_REQUIRE(k == 2 * stride && pad == (k - stride) / 2 && dil == 1, "test_conv1d: transposed needs k=2s, pad=(k-s)/2");
            rows = stride * Cout;
            BM = kx::conv_pick_bm(rows);
            packed = dm.get<float>(kx::packed_conv_floats(rows, Cin, 2, BM));
            kx::launch_pack_convT(dw, packed, Cin, Cout, stride, BM, nullptr);
            a.K = 2;
            a.stride = 1;
            a.pad = 1;
            a.dil = 1;
            a.store = kx::ST_UPSCATTER;
            a.up_s = stride;
            a.up_pad = pad;
            a.up_cout = Cout;
            a.up_off = ex.up_off;
            a.up_reflect = ex.up_reflect;
        }
        a.Cout = rows;
        a.w = packed;
        a.bias = bias ? dm.up(bias, (size_t)Cout) : nullptr;
        if (norm) {
            const float* dn = dm.up(norm, (size_t)3 * B * Cin);
            a.nmean = dn;
            a.nscale = dn + (size_t)B * Cin;
            a.nshift = dn + (size_t)2 * B * Cin;
            a.n_bs = Cin;
        }
        a.act = act;
        a.slope = slope;
        a.alpha = alpha ? dm.up(alpha, (size_t)Cin) : nullptr;
        KX_REQUIRE(act != kx::ACT_SNAKE || alpha, "test_conv1d: snake needs alpha");
        float* dy = dm.get<float>((size_t)B * Cout * y_ld);
        if (ex.pad_ld) {  // y holds the running sum (accumulate) or zeros; the row padding holds a sentinel nobody may touch
            std::vector<float> y0((size_t)B * Cout * Lout, 0.f);
            const std::vector<float> yp = padded(ex.accum ? y : y0.data(), B * Cout, Lout, y_ld, sentinel);
            KX_HIP(hipMemcpy(dy, yp.data(), yp.size() * 4, hipMemcpyHostToDevice));
        } else if (ex.accum)
            KX_HIP(hipMemcpy(dy, y, (size_t)B * Cout * Lout * 4, hipMemcpyHostToDevice));  // y holds the running sum
        else
            KX_HIP(hipMemset(dy, 0, (size_t)B * Cout * Ly * 4));
        a.y = dy;
        a.y_bs = (long)Cout * y_ld;
        a.y_ld = y_ld;
        a.out_mul = ex.out_mul;
        a.out_div = ex.out_div;
        a.accum = ex.accum;
        if (ex.resid) {
            KX_REQUIRE(!transposed || !ex.pad_ld, "test_conv1d: residual of a transposed conv on dense rows only");
            if (ex.pad_ld) {
                const std::vector<float> rp = padded(ex.resid, B * Cout, Lout, y_ld, poison);
                a.resid = dm.up(rp.data(), rp.size());
            } else
                a.resid = dm.up(ex.resid, (size_t)B * Cout * Ly);
            a.r_bs = (long)Cout * y_ld;
            a.r_ld = y_ld;
        }
        float2* d_part = nullptr;
        int cols_per_tile = 0;
        if (ex.stats_out) {
            KX_REQUIRE(!transposed && !ex.accum, "test_conv1d: fused statistics come with plain, non-accumulating stores");
            int bn, wn;
            if (mode != kx::CONV_F32) {
                kx::conv16_pick_tile(BM, Lout, B, rows, a.K, a.dil, a.stride, &bn, &wn, a.ws_force, true, a.act, (Cin + 15) / 16, f8 ? 2 : 0);
            } else {
                bn = kx::conv_bn(BM);
                wn = BM == 128 ? 2 : 4;
            }
            a.stat_tiles = ((Lout + bn - 1) / bn) * wn;
            cols_per_tile = bn / wn;
            d_part = dm.get<float2>((size_t)B * rows * a.stat_tiles);
            KX_HIP(hipMemset(d_part, 0, (size_t)B * rows * a.stat_tiles * sizeof(float2)));
            a.stat_part = d_part;
        }
        if (mode != kx::CONV_F32) {
            const float amax = kx::device_absmax(dw, (long)Cout * Cin * k, nullptr);
            const int ws = kx::pick_weight_shift(amax);
            const int Kp = transposed ? 2 : k;
            void* p16 = dm.get<unsigned short>(kx::packed_conv16_halves(rows, Cin, Kp, BM));
            if (transposed)
                kx::launch_pack_convT16(dw, p16, Cin, Cout, stride, BM, std::ldexp(1.0f, ws), nullptr);
            else {
                kx::PackSrc src{{dw, nullptr, nullptr}, {Cout, 0, 0}};
                kx::launch_pack_conv16(src, p16, Cout, Cin, k, BM, std::ldexp(1.0f, ws), nullptr);
            }
            a.w16 = p16;
            if (f8) {
                KX_REQUIRE(!transposed && BM == 128, "test_conv1d: f16f8 images exist for 128-row tiles of plain convs");
                void* p8 = dm.get<unsigned char>(kx::packed_conv8x_bytes(rows, Cin, k));
                kx::launch_pack_conv8x(p16, p8, rows, Cin, k, nullptr);
                a.w8x = p8;
            }
            a.n_chunks16 = (Cin + 15) / 16;
            a.w_unscale = std::ldexp(1.0f, -ws);
            a.x_prescale = 1.0f;
            a.xcd_swizzle = 1;
            const int max_cols = transposed ? L + 1 : Lout;
            if (ex.flat && B > 1) {
                const int fbn = kx::conv16_flat_bn(a, BM, B, max_cols);
                if (fbn) {
                    const kx::LenMap lm = transposed ? a.in_len : a.out_len;
                    int* d_pre = dm.get<int>((size_t)B + 1);
                    kx::launch_tile_prefix(lm, transposed ? 1 : 0, fbn, B, d_pre, nullptr);
                    int total = 0;
                    for (int b = 0; b < B; ++b) {
                        const int cols = lens[b] * lm.mul + lm.add + (transposed ? 1 : 0);
                        total += cols > 0 ? (cols + fbn - 1) / fbn : 0;
                    }
                    a.tile_prefix = d_pre;
                    a.flat_ny = (rows + 127) / 128;
                    a.flat_B = B;
                    a.flat_tiles_host = total;
                    a.flat_bn_host = fbn;
                }
            }
            if (pre) {
                KX_REQUIRE(BM == 128 && a.K >= 2 && a.act != kx::ACT_SNAKE && kx::conv16_da_eligible(BM, a.K, a.dil, a.stride, 0),
                           "test_conv1d: this layer has no pre-split form");
                const long img_bs = (long)kx::conv16_pre_image_bytes(Cin, x_ld);
                unsigned char* img = dm.get<unsigned char>((size_t)B * img_bs);
                KX_HIP(hipMemset(img, 0xff, (size_t)B * img_bs));  // (NaN halves wherever the pass does not write)
                kx::launch_split_image(a, B, L, img, img_bs, nullptr);
                a.x16 = img;
                a.x16_bs = img_bs;
                a.x16_ld = x_ld;
            }
            kx::launch_conv1d_f16x3(a, BM, B, max_cols, nullptr);
        } else {
            kx::launch_conv1d(a, BM, B, transposed ? L + 1 : Lout, nullptr);
        }
        KX_HIP(hipDeviceSynchronize());
        if (ex.pad_ld) {
            std::vector<float> yp((size_t)B * Cout * y_ld);
            KX_HIP(hipMemcpy(yp.data(), dy, yp.size() * 4, hipMemcpyDeviceToHost));
            for (int r = 0; r < B * Cout; ++r) {
                std::memcpy(y + (size_t)r * Lout, &yp[(size_t)r * y_ld], (size_t)Lout * 4);
                for (int c = Lout; c < y_ld; ++c)
                    if (yp[(size_t)r * y_ld + c] != sentinel) throw Error(KX_ERR_STATE, "test_conv1d: the kernel wrote into the row padding");
            }
        } else
            KX_HIP(hipMemcpy(y, dy, (size_t)B * Cout * Ly * 4, hipMemcpyDeviceToHost));
        if (ex.stats_out) {
            std::vector<float2> part((size_t)B * rows * a.stat_tiles);
            KX_HIP(hipMemcpy(part.data(), d_part, part.size() * sizeof(float2), hipMemcpyDeviceToHost));
            const int used = (Lout + cols_per_tile - 1) / cols_per_tile;
            for (size_t br = 0; br < (size_t)B * rows; ++br) {
                double sm = 0.0, sq = 0.0;
                for (int t = 0; t < used && t < a.stat_tiles; ++t) {
                    sm += part[br * a.stat_tiles + t].x;
                    sq += part[br * a.stat_tiles + t].y;
                }
                ex.stats_out[2 * br] = (float)sm;
                ex.stats_out[2 * br + 1] = (float)sq;
            }
        }
    });
}

int kx_test_conv1d(int device_id, const float* x, int B, int Cin, int L, const float* w, const float* bias, int Cout,
                   int k, int stride, int pad, int dil, int transposed, int act, float slope, const float* alpha,
                   const float* norm, float* y, int Lout, int mode, char* err, size_t err_len) {
    return test_conv1d_impl(device_id, x, B, Cin, L, w, bias, Cout, k, stride, pad, dil, transposed, act, slope, alpha, norm,
                            y, Lout, mode, ConvTestExtra{}, err, err_len);
}

int kx_test_conv_transpose(int device_id, const float* x, int B, int Cin, int L, const float* w, const float* bias, int Cout,
                           int stride, int act, float slope, const float* resid, int up_off, float* y, int mode, char* err,
                           size_t err_len) {
    ConvTestExtra ex;
    ex.resid = resid;
    ex.up_off = up_off ? 1 : 0;
    ex.up_reflect = up_off ? 1 : 0;
    const int k = 2 * stride, pad = (k - stride) / 2;
    const int Lout = (L - 1) * stride - 2 * pad + k;
    return test_conv1d_impl(device_id, x, B, Cin, L, w, bias, Cout, k, stride, pad, 1, 1, act, slope, nullptr, nullptr, y, Lout, mode, ex,
                            err, err_len);
}

int kx_test_conv1d_epilogue(int device_id, const float* x, int B, int Cin, int L, const float* w, const float* bias,
                            int Cout, int k, int pad, int dil, const float* resid, int accumulate, float out_mul,
                            float out_div, float* y, float* stats_out, int mode, char* err, size_t err_len) {
    ConvTestExtra ex;
    ex.resid = resid;
    ex.accum = accumulate;
    ex.out_mul = out_mul;
    ex.out_div = out_div;
    ex.stats_out = stats_out;
    const int Lout = L + 2 * pad - dil * (k - 1);
    return test_conv1d_impl(device_id, x, B, Cin, L, w, bias, Cout, k, 1, pad, dil, 0, 0, 0.f, nullptr, nullptr, y, Lout, mode,
                            ex, err, err_len);
}

int kx_test_conv1d_full(int device_id, const float* x, int B, int Cin, int L, const int32_t* lens, int pad_ld, const float* w,
                        const float* bias, int Cout, int k, int pad, int dil, int act, float slope, const float* alpha,
                        const float* norm, const float* resid, int accumulate, float out_mul, float out_div, float* y,
                        float* stats_out, int mode, char* err, size_t err_len) {
    ConvTestExtra ex;
    ex.resid = resid;
    ex.accum = accumulate;
    ex.out_mul = out_mul;
    ex.out_div = out_div;
    ex.stats_out = stats_out;
    ex.lens = lens;
    ex.pad_ld = pad_ld & 1;
    ex.flat = (pad_ld >> 1) & 1;
    const int Lout = L + 2 * pad - dil * (k - 1);
    return test_conv1d_impl(device_id, x, B, Cin, L, w, bias, Cout, k, 1, pad, dil, 0, act, slope, alpha, norm, y, Lout, mode, ex,
                            err, err_len);
}

int kx_test_lstm(int device_id, const float* x, int B, int L, int n_in, const float* w_ih, const float* w_hh,
                 const float* b_ih, const float* b_hh, const float* w_ih_r, const float* w_hh_r, const float* b_ih_r,
                 const float* b_hh_r, float* y, char* err, size_t err_len) {
    return guarded_free(err, err_len, [&] {
        check_device(device_id);
        KX_REQUIRE(x && y && B > 0 && L > 0 && n_in > 0, "test_lstm: bad argument");
        KX_HIP(hipSetDevice(device_id));
        DevMem dm;
        std::vector<float> xc((size_t)B * n_in * L);  // [B,L,n_in] -> channel-major [B][n_in][L]
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < L; ++t)
                for (int c = 0; c < n_in; ++c) xc[((size_t)b * n_in + c) * L + t] = x[((size_t)b * L + t) * n_in + c];
        std::vector<int> lens(B, L);
        int* d_len = dm.up(lens.data(), B);
        kx::ConvArgs a{};
        a.x = dm.up(xc.data(), xc.size());
        a.x_bs = (long)n_in * L;
        a.x_ld = L;
        a.Cin = n_in;
        a.in_len = kx::LenMap{d_len, 1, 0};
        a.out_len = a.in_len;
        a.n_chunks = (n_in + kx::CONV_CK - 1) / kx::CONV_CK;
        float* packed = dm.get<float>(kx::packed_conv_floats(2048, n_in, 1, 128));
        kx::PackSrc src{{dm.up(w_ih, (size_t)1024 * n_in), dm.up(w_ih_r, (size_t)1024 * n_in), nullptr}, {1024, 1024, 0}};
        kx::launch_pack_conv(src, packed, 2048, n_in, 1, 128, nullptr);
        float* bias = dm.get<float>(2048);
        kx::launch_vec_add(dm.up(b_ih, 1024), dm.up(b_hh, 1024), bias, 1024, nullptr);
        kx::launch_vec_add(dm.up(b_ih_r, 1024), dm.up(b_hh_r, 1024), bias + 1024, 1024, nullptr);
        float* whhT = dm.get<float>(2 * 256 * 1024);
        kx::launch_transpose_whh(dm.up(w_hh, 1024 * 256), whhT, nullptr);
        kx::launch_transpose_whh(dm.up(w_hh_r, 1024 * 256), whhT + 256 * 1024, nullptr);
        float* gx = dm.get<float>((size_t)B * L * 2048);
        a.w = packed;
        a.bias = bias;
        a.K = 1; a.dil = 1; a.stride = 1; a.pad = 0;
        a.Cout = 2048;
        a.y = gx;
        a.y_bs = (long)L * 2048;
        a.y_ld = 2048;
        a.out_mul = 1.f; a.out_div = 1.f;
        a.store = kx::ST_TMAJOR;
        a.up_cout = 1;
        kx::launch_conv1d(a, 128, B, L, nullptr);
        float* dy = dm.get<float>((size_t)B * 512 * L);
        // (the hook runs the product's two-CU recurrence: exchange buffer + sticky error word as Model holds them)
        unsigned long long* xchg = dm.get<unsigned long long>(kx::lstm_exchange_bytes(B) / sizeof(unsigned long long));
        KX_HIP(hipMemset(xchg, 0, kx::lstm_exchange_bytes(B)));
        unsigned* errw = dm.get<unsigned>(1);
        KX_HIP(hipMemset(errw, 0, sizeof(unsigned)));
        kx::launch_lstm(gx, (long)L * 2048, 2048, whhT, dy, (long)512 * L, L, kx::LenMap{d_len, 1, 0}, B, xchg, errw, nullptr);
        KX_HIP(hipDeviceSynchronize());
        unsigned herr = 0;
        KX_HIP(hipMemcpy(&herr, errw, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (herr) throw Error(KX_ERR_DEVICE, "test_lstm: the two-CU recurrence timed out waiting for its partner");
        KX_HIP(hipDeviceSynchronize());
        std::vector<float> yc((size_t)B * 512 * L);
        KX_HIP(hipMemcpy(yc.data(), dy, yc.size() * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < 512; ++c)
                for (int t = 0; t < L; ++t) y[((size_t)b * L + t) * 512 + c] = yc[((size_t)b * 512 + c) * L + t];
    });
}

int kx_test_attention(int device_id, const float* qkv, const int32_t* lens, int B, int T, float* ctx, char* err,
                      size_t err_len) {
    return guarded_free(err, err_len, [&] {
        check_device(device_id);
        KX_REQUIRE(qkv && lens && ctx && B > 0 && T > 0 && T <= 512, "test_attention: bad argument");
        for (int b = 0; b < B; ++b) KX_REQUIRE(lens[b] >= 1 && lens[b] <= T, "test_attention: lens out of range");
        KX_HIP(hipSetDevice(device_id));
        DevMem dm;
        const int ld = (T + 31) & ~31;  // rows padded as in the model (128-byte lines)
        std::vector<float> q((size_t)B * 2304 * ld, std::nanf(""));  // padding holds NaNs: the kernel must not use it
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < 2304; ++c)
                std::memcpy(&q[((size_t)b * 2304 + c) * ld], &qkv[((size_t)b * 2304 + c) * T], (size_t)T * 4);
        const float* d_q = dm.up(q.data(), q.size());
        const int* d_len = dm.up(lens, B);
        float* d_ctx = dm.get<float>((size_t)B * 768 * ld);
        KX_HIP(hipMemset(d_ctx, 0, (size_t)B * 768 * ld * 4));
        kx::launch_attention(d_q, (long)2304 * ld, ld, d_ctx, (long)768 * ld, ld, d_len, B, T, nullptr);
        KX_HIP(hipDeviceSynchronize());
        std::vector<float> c((size_t)B * 768 * ld);
        KX_HIP(hipMemcpy(c.data(), d_ctx, c.size() * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b)
            for (int r = 0; r < 768; ++r)
                std::memcpy(&ctx[((size_t)b * 768 + r) * T], &c[((size_t)b * 768 + r) * ld], (size_t)T * 4);
    });
}

int kx_test_source(int device_id, const float* f0, int B, int F2, const float* lin_w, float lin_b, uint64_t seed,
                   uint64_t utt_base, int noise_off, float* out, char* err, size_t err_len) {
    return guarded_free(err, err_len, [&] {
        check_device(device_id);
        KX_REQUIRE(f0 && lin_w && out && B > 0 && F2 > 0 && (F2 % 2) == 0, "test_source: bad argument");
        KX_HIP(hipSetDevice(device_id));
        kx::init_dft_tables();
        DevMem dm;
        const int F = F2 / 2;
        std::vector<int> fr(B, F);
        int* d_fr = dm.up(fr.data(), B);
        const float* d_f0 = dm.up(f0, (size_t)B * F2);
        const float* d_w = dm.up(lin_w, 9);
        const float* d_b = dm.up(&lin_b, 1);
        float* phase = dm.get<float>((size_t)B * 9 * F2);
        float* har = dm.get<float>((size_t)B * 600 * F);
        kx::launch_source(d_f0, F2, d_fr, B, F, d_w, d_b, seed, utt_base, nullptr, noise_off, phase, har, (long)600 * F, nullptr);
        KX_HIP(hipDeviceSynchronize());
        KX_HIP(hipMemcpy(out, har, (size_t)B * 600 * F * 4, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
