// Host side of the MI355X Kokoro-82M forward: weight registry + repacking, arenas and the
// launch sequence that replaces `sess.run` (kokorox/src/onn/ort_koko.rs:79).  The graph is
// the published Kokoro-82M; stage comments name the upstream module.
#include <chrono>
#include "model.h"

#include <sys/stat.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>

#include "onnx_import.h"

namespace kx {

static constexpr float RSQRT2 = 0.70710678118654752f;
// row strides are multiples of 32 floats: every 32-column half-wave store of the conv epilogue is one whole
// 128-byte line (arenas are 256-byte aligned)
static inline int up4(int x) { return (x + 31) & ~31; }

// CUs of the launches this thread is issuing: the device's, or the model's share of them (CU-partitioned models)
static thread_local int tl_cu_override = 0;
int cu_count_override() { return tl_cu_override; }
struct CuScope {
    int saved;
    explicit CuScope(int n) : saved(tl_cu_override) { tl_cu_override = n; }
    ~CuScope() { tl_cu_override = saved; }
};

Model::Model(int dev, int part, int n_parts) : device(dev), part_(part), n_parts_(n_parts) {
    if (const char* e = getenv("KOKOROX_CONV"))
        conv_mode = (strcmp(e, "f32") == 0) ? CONV_F32 : (strcmp(e, "f16") == 0 ? CONV_F16 : (strcmp(e, "bf16") == 0 ? CONV_BF16 : (strcmp(e, "f16f8") == 0 ? CONV_F16F8 : CONV_F16X3)));
    if (const char* e = getenv("KOKOROX_STFT")) stft_variant = (strcmp(e, "torch") == 0) ? STFT_TORCH : STFT_ONNX;
    // (per-device state of the library -- the per-device turn of the forwards, dynamic-LDS attribute limits, CU counts -- is
    // kept in tables of KX_MAX_DEVICES entries: an id beyond them is refused here, never aliased onto another device's entry)
    if (dev < 0 || dev >= KX_MAX_DEVICES) throw Error(1, "device id outside 0.." + std::to_string(KX_MAX_DEVICES - 1));
    KX_HIP(hipSetDevice(device));
    // Non-blocking: the model's streams never synchronise with the legacy null stream, so nothing one model does can stall
    // another model on the same GPU (two models per GPU in the server, kx_create_replicas with repeated ids).  Everything
    // on the forward's path is issued on stream_ or ordered to it by events; kx_infer_device orders its device inputs
    // after the caller's null-stream work explicitly (infer_device_after_null).
    // A CU-partitioned model (kx_create_partition; kx_create_replicas with a device id given k times): every stream of the model
    // is confined to its share of the CUs -- bits [part, part + 1) x CUs / n_parts of the queue's CU mask, which the driver deals
    // over the XCCs (probed: either half of the bits is 128 CUs on all 8 XCCs) -- so k models run their forwards side by side
    // on one GPU without ever competing for a CU: a recurrence's workgroups, which need a whole CU each, cannot starve behind
    // another model's conv workgroups (profiles/r04_serve_models_per_gpu.txt), and the under-filled phases of one forward (the
    // token-axis front half, the recurrences) run beside the other's convs.  (Such streams synchronise with the legacy null
    // stream -- the extension has no flags argument; nothing on the forward's path touches that stream.)
    KX_REQUIRE(n_parts >= 1 && n_parts <= 8 && part >= 0 && part < n_parts, "model: partition must be 0 .. n_parts - 1 of 1 .. 8");
    if (n_parts > 1) {
        hipDeviceProp_t prop;
        KX_HIP(hipGetDeviceProperties(&prop, device));
        const int cus = prop.multiProcessorCount;
        const int lo = (int)((long)cus * part / n_parts), hi = (int)((long)cus * (part + 1) / n_parts);
        KX_REQUIRE(hi - lo >= 16, "model: a partition needs at least 16 CUs");
        cu_mask_.assign((size_t)(cus + 31) / 32, 0u);
        for (int i = lo; i < hi; ++i) cu_mask_[(size_t)i >> 5] |= 1u << (i & 31);
        cu_count_ = hi - lo;
    }
    new_stream(&stream_);
    main_stream_ = stream_;
    KX_HIP(hipMalloc((void**)&d_dev_err_, sizeof(unsigned)));
    KX_HIP(hipMemsetAsync(d_dev_err_, 0, sizeof(unsigned), stream_));
    KX_HIP(hipHostMalloc((void**)&h_words_, 8 * sizeof(unsigned), hipHostMallocDefault));
    memset(h_words_, 0, 8 * sizeof(unsigned));
    KX_HIP(hipEventCreateWithFlags(&ev_null_, hipEventDisableTiming));
    KX_HIP(hipStreamSynchronize(stream_));
    new_stream(&stream2_);
    lanes_[0].stream = stream_;
    for (int i = 1; i < N_LANES; ++i) new_stream(&lanes_[i].stream);
    if (const char* e = getenv("KX_LANES")) set_lanes(atoi(e));
    KX_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    KX_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    init_dft_tables();
}

// (CU-masked streams are created and retired one at a time, process-wide: kx_create_replicas builds its models on one thread each,
// and the runtime of this image has shown one stall in masked-stream teardown already -- see ~Model)
static std::mutex& masked_stream_mutex() {
    static std::mutex m;
    return m;
}

void Model::new_stream(hipStream_t* s) {
    if (cu_mask_.empty()) {
        KX_HIP(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
        return;
    }
    std::lock_guard<std::mutex> lk(masked_stream_mutex());
    KX_HIP(hipExtStreamCreateWithCUMask(s, (uint32_t)cu_mask_.size(), cu_mask_.data()));
}

Model::~Model() {
    const bool tr = getenv("KX_TRACE_DTOR") != nullptr;
    // CU-masked streams: with the ROCm 7.2 runtime of this image (not with the one torch bundles) hipStreamDestroy of the SECOND
    // masked stream of a device hung for good although every stream had been synchronised on its own; behind one
    // hipDeviceSynchronize it returns at once (probed: tools/_dbg in git history, profiles/r05_experiments_not_kept.txt item 9).
    // KX_MASKED_DESTROY=0 leaves the masked streams to the runtime's own teardown instead (also probed: clean exit).
    // One stall of the C++ replicas demo in 20 runs of the suite was seen AFTER that fix (the stage was not recorded; the demo and
    // this destructor now report theirs): masked streams are retired one model at a time under a process-wide lock.
    const int md = getenv("KX_MASKED_DESTROY") ? atoi(getenv("KX_MASKED_DESTROY")) : 2;
    const bool masked = !cu_mask_.empty();
    auto T = [&](const char* what) { if (tr) { fprintf(stderr, "dtor: %s\n", what); fflush(stderr); } };
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> masked_lk;
    if (masked) masked_lk = std::unique_lock<std::mutex>(masked_stream_mutex());
    T("device sync");
    if (masked && md == 2) (void)hipDeviceSynchronize();
    auto destroy = [&](hipStream_t st) { if (!masked || md >= 1) (void)hipStreamDestroy(st); };
    T("sync main");
    if (stream_) (void)hipStreamSynchronize(stream_);
    T("sync side");
    if (stream2_) (void)hipStreamSynchronize(stream2_);
    for (int i = 1; i < N_LANES; ++i)
        if (lanes_[i].stream) {
            T("sync lane");
            (void)hipStreamSynchronize(lanes_[i].stream);
            T("destroy lane");
            destroy(lanes_[i].stream);
        }
    T("lanes gone");
    for (hipEvent_t e : lane_ev_) (void)hipEventDestroy(e);
    for (void* p : owned_) (void)hipFree(p);
    if (d_dev_err_) (void)hipFree(d_dev_err_);
    if (h_words_) (void)hipHostFree(h_words_);
    if (h_stage_) (void)hipHostFree(h_stage_);
    if (ev_null_) (void)hipEventDestroy(ev_null_);
    if (ev_done_) (void)hipEventDestroy(ev_done_);
    for (auto* p : d_xchg_)
        if (p) (void)hipFree(p);
    for (Arena* a : {&arenaT_, &arenaF_, &arenaIO_})
        if (a->base) (void)hipFree(a->base);
    if (blob_) (void)hipFree(blob_);
    for (hipEvent_t e : ev_) (void)hipEventDestroy(e);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
    T("destroy side");
    if (stream2_) destroy(stream2_);
    T("destroy main");
    if (stream_) destroy(stream_);
    T("done");
}

// ---- weight container ------------------------------------------------------------------------
static void parse_table(const unsigned char* hdr, size_t hdr_bytes, size_t total,
                        std::map<std::string, TensorInfo>& table) {
    uint32_t n;
    memcpy(&n, hdr + 8, 4);
    if (64 + (size_t)n * 128 > hdr_bytes) throw Error(2, "weight blob: truncated tensor table");
    for (uint32_t i = 0; i < n; ++i) {
        const unsigned char* e = hdr + 64 + (size_t)i * 128;
        char name[89];
        memcpy(name, e, 88);
        name[88] = 0;
        uint32_t dt, nd, dims[4];
        uint64_t off, nb;
        memcpy(&dt, e + 88, 4);
        memcpy(&nd, e + 92, 4);
        memcpy(dims, e + 96, 16);
        memcpy(&off, e + 112, 8);
        memcpy(&nb, e + 120, 8);
        if (dt != 0 || nd > 4 || off + nb > total || (off & 255)) throw Error(2, std::string("weight blob: bad entry ") + name);
        TensorInfo ti;
        ti.offset = off;
        ti.nbytes = nb;
        ti.ndim = (int)nd;
        size_t cnt = 1;
        for (int k = 0; k < (int)nd; ++k) {
            ti.dims[k] = (int)dims[k];
            cnt *= dims[k];
        }
        if (cnt * 4 != nb) throw Error(2, std::string("weight blob: size mismatch ") + name);
        table[name] = ti;
    }
}

static size_t check_header(const unsigned char* h, size_t have) {
    if (have < 64 || memcmp(h, "KXHIPW01", 8) != 0) throw Error(2, "weight blob: bad magic (expected KXHIPW01)");
    uint64_t total;
    memcpy(&total, h + 24, 8);
    return (size_t)total;
}

static std::vector<unsigned char> read_all(const char* path, const char* what) {
    FILE* f = fopen(path, "rb");
    if (!f) throw Error(2, std::string("cannot open ") + what + ": " + path);
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0) {
        fclose(f);
        throw Error(2, std::string("cannot size ") + what + ": " + path);
    }
    std::vector<unsigned char> host((size_t)sz);
    const size_t got = host.empty() ? 0 : fread(host.data(), 1, host.size(), f);
    fclose(f);
    if (got != host.size()) throw Error(2, std::string("short read on ") + what + ": " + path);
    return host;
}

std::vector<unsigned char> import_onnx_bytes(const unsigned char* data, size_t n, int* variant) {
    try {
        ImportInfo info;
        std::vector<unsigned char> blob = onnx_to_kxw(data, n, &info);
        if (variant) *variant = info.variant();
        return blob;
    } catch (const ImportError& e) {
        throw Error(2, std::string("weight file is not a KXHIPW01 blob (bad magic) and not a readable ONNX model: ") + e.what());
    }
}

// The KXHIPW01 image behind `path`.  The path is either the library's own container or — what the reference passes to
// OrtKoko::new (koko.rs:570-573, hf_cache.rs:128-158) — the `.onnx` file, which is converted in memory
// (onnx_import.cpp).  `<path>.kxw` beside an .onnx is used instead when it is at least as new as the .onnx and whole;
// it is written only when KOKOROX_KXW_CACHE=1 (a library should not drop files into a model cache unasked).
std::vector<unsigned char> read_weight_file(const char* path, int* variant) {
    KX_REQUIRE(path && *path, "kx_create: empty weights path");
    if (variant) *variant = 0;
    std::vector<unsigned char> host = read_all(path, "weight file");
    if (is_kxw_magic(host.data(), host.size())) {
        const size_t total = check_header(host.data(), host.size());
        if (total != host.size()) throw Error(2, "weight blob: file size does not match header");
        return host;
    }
    // The converted image may be cached beside the source -- ONLY when KOKOROX_KXW_CACHE=1 says so (a file the library never wrote
    // is never trusted), and only while the cache's stamp names exactly this source: its size, its modification time to the
    // nanosecond and the importer's version (cp -p / mv / a re-pointed Hugging Face blob symlink keep an older mtime: a "not
    // older than the source" test would load the former variant's weights).
    const std::string cache = std::string(path) + ".kxw", stamp_path = cache + ".src";
    const char* ce = getenv("KOKOROX_KXW_CACHE");
    const bool use_cache = ce && strcmp(ce, "1") == 0;
    struct stat so;
    std::string stamp;
    if (use_cache && stat(path, &so) == 0) {
        stamp = "kxw-cache 1 importer " + std::to_string(KX_IMPORTER_VERSION) + " size " + std::to_string((long long)so.st_size) + " mtime " +
                std::to_string((long long)so.st_mtim.tv_sec) + "." + std::to_string((long)so.st_mtim.tv_nsec) + "\n";
        try {
            const std::vector<unsigned char> st = read_all(stamp_path.c_str(), "weight cache stamp");
            if (std::string(st.begin(), st.end()) == stamp) {
                std::vector<unsigned char> c = read_all(cache.c_str(), "weight cache");
                if (is_kxw_magic(c.data(), c.size()) && check_header(c.data(), c.size()) == c.size()) {
                    if (variant) *variant = -1;  // (a cached conversion: the source's kind was not looked at again)
                    return c;
                }
            }
        } catch (const Error&) {  // no cache, or an unreadable one, is not an error: convert
        }
    }
    int var = 1;
    std::vector<unsigned char> blob = import_onnx_bytes(host.data(), host.size(), &var);
    if (variant) *variant = var;
    if (var >= 3)
        fprintf(stderr,
                "kokorox-hip: %s is a %d-bit quantised ONNX variant: its weights are de-quantised at load and the model runs f32-class "
                "arithmetic, which is NOT what ONNX Runtime computes for this file (it quantises the activations at run time: "
                "DynamicQuantizeLinear -> MatMulInteger / ConvInteger / MatMulNBits).  Parity with the reference is claimed for "
                "onnx/model.onnx only.\n",
                path, var == 3 ? 8 : 4);
    if (use_cache && !stamp.empty()) {
        const std::string tmp = cache + ".tmp." + std::to_string((long)getpid());
        if (FILE* f = fopen(tmp.c_str(), "wb")) {
            const bool ok = fwrite(blob.data(), 1, blob.size(), f) == blob.size();
            if (fclose(f) != 0 || !ok || rename(tmp.c_str(), cache.c_str()) != 0) (void)remove(tmp.c_str());
            else if (FILE* g = fopen((stamp_path + ".tmp").c_str(), "wb")) {  // (the stamp last: a cache without it is ignored)
                const bool ok2 = fwrite(stamp.data(), 1, stamp.size(), g) == stamp.size();
                if (fclose(g) != 0 || !ok2 || rename((stamp_path + ".tmp").c_str(), stamp_path.c_str()) != 0) (void)remove((stamp_path + ".tmp").c_str());
            }
        }
    }
    return blob;
}

void Model::load_file(const char* path) {
    const std::vector<unsigned char> host = read_weight_file(path, &source_variant_);
    const size_t n = host.size();
    parse_table(host.data(), n, n, table_);
    KX_HIP(hipSetDevice(device));
    KX_HIP(hipMalloc((void**)&blob_, n));
    blob_bytes_ = n;
    KX_HIP(hipMemcpy(blob_, host.data(), n, hipMemcpyHostToDevice));
    build();
}

void Model::load_device_blob(const void* d_blob, size_t n, bool adopt) {
    KX_REQUIRE(d_blob && n >= 64, "kx_create_from_device_blob: empty blob");
    KX_HIP(hipSetDevice(device));
    unsigned char h64[64];
    KX_HIP(hipMemcpy(h64, d_blob, 64, hipMemcpyDeviceToHost));
    const size_t total = check_header(h64, 64);
    if (total != n) throw Error(2, "weight blob: size does not match header");
    uint32_t nt;
    memcpy(&nt, h64 + 8, 4);
    const size_t hdr_bytes = 64 + (size_t)nt * 128;
    if (hdr_bytes > n) throw Error(2, "weight blob: truncated tensor table");
    std::vector<unsigned char> hdr(hdr_bytes);
    KX_HIP(hipMemcpy(hdr.data(), d_blob, hdr_bytes, hipMemcpyDeviceToHost));
    parse_table(hdr.data(), hdr_bytes, total, table_);
    blob_bytes_ = n;
    if (adopt) {  // the caller hands over a hipMalloc'd blob on this device (kx_create_replicas): no second copy
        blob_ = static_cast<char*>(const_cast<void*>(d_blob));
    } else {
        KX_HIP(hipMalloc((void**)&blob_, n));
        KX_HIP(hipMemcpy(blob_, d_blob, n, hipMemcpyDeviceToDevice));
        KX_HIP(hipStreamSynchronize(nullptr));  // (a device-to-device hipMemcpy may return before it has run)
    }
    build();
}

const TensorInfo& Model::info(const std::string& name) const {
    auto it = table_.find(name);
    if (it == table_.end()) throw Error(2, "weight blob: missing tensor " + name);
    return it->second;
}
const float* Model::wt(const std::string& name) const {
    return reinterpret_cast<const float*>(blob_ + info(name).offset);
}
float* Model::dev_alloc(size_t floats) {
    void* p = nullptr;
    KX_HIP(hipMalloc(&p, floats * sizeof(float)));
    owned_.push_back(p);
    return static_cast<float*>(p);
}

// split-f16 image of the same weights (normal convs: src; transposed: wT [Cin][Cout][2s])
void Model::pack16(ConvW& c, const PackSrc& src, const float* wT, int n_src_floats[3]) {
    float amax = 0.f;
    if (wT) {
        amax = device_absmax(wT, n_src_floats[0], stream_);
    } else {
        for (int i = 0; i < 3; ++i)
            if (src.p[i]) amax = std::fmax(amax, device_absmax(src.p[i], n_src_floats[i], stream_));
    }
    const int ws = pick_weight_shift(amax);
    c.unscale = std::ldexp(1.0f, -ws);
    c.n_chunks16 = (c.Cin + 15) / 16;
    const size_t halves = packed_conv16_halves(c.rows, c.Cin, c.K, c.BM);
    void* p = dev_alloc((halves + 1) / 2);
    if (wT)
        launch_pack_convT16(wT, p, c.Cin, c.up_cout, c.up_s, c.BM, std::ldexp(1.0f, ws), stream_);
    else
        launch_pack_conv16(src, p, c.rows, c.Cin, c.K, c.BM, std::ldexp(1.0f, ws), stream_);
    c.w16 = p;
}

ConvW Model::make_conv(const std::string& name, bool bias) {
    const TensorInfo& ti = info(name + ".weight");
    ConvW c;
    c.rows = ti.dims[0];
    c.Cin = ti.dims[1];
    c.K = ti.ndim >= 3 ? ti.dims[2] : 1;
    c.BM = conv_pick_bm(c.rows);
    c.n_chunks = (c.Cin + CONV_CK - 1) / CONV_CK;
    float* p = dev_alloc(packed_conv_floats(c.rows, c.Cin, c.K, c.BM));
    PackSrc src{{wt(name + ".weight"), nullptr, nullptr}, {c.rows, 0, 0}};
    launch_pack_conv(src, p, c.rows, c.Cin, c.K, c.BM, stream_);
    c.w = p;
    c.bias = (bias && has(name + ".bias")) ? wt(name + ".bias") : nullptr;
    int nf[3] = {c.rows * c.Cin * c.K, 0, 0};
    pack16(c, src, nullptr, nf);
    return c;
}

ConvW Model::make_conv_cat(const std::vector<std::string>& names) {
    KX_REQUIRE(names.size() <= 3 && !names.empty(), "make_conv_cat");
    ConvW c;
    PackSrc src{{nullptr, nullptr, nullptr}, {0, 0, 0}};
    for (size_t i = 0; i < names.size(); ++i) {
        const TensorInfo& ti = info(names[i] + ".weight");
        src.p[i] = wt(names[i] + ".weight");
        src.rows[i] = ti.dims[0];
        c.rows += ti.dims[0];
        c.Cin = ti.dims[1];
        c.K = ti.ndim >= 3 ? ti.dims[2] : 1;
    }
    c.BM = conv_pick_bm(c.rows);
    c.n_chunks = (c.Cin + CONV_CK - 1) / CONV_CK;
    float* p = dev_alloc(packed_conv_floats(c.rows, c.Cin, c.K, c.BM));
    launch_pack_conv(src, p, c.rows, c.Cin, c.K, c.BM, stream_);
    c.w = p;
    int nf[3] = {src.rows[0] * c.Cin * c.K, src.rows[1] * c.Cin * c.K, src.rows[2] * c.Cin * c.K};
    pack16(c, src, nullptr, nf);
    float* bias = dev_alloc(c.rows);
    int r0 = 0;
    for (size_t i = 0; i < names.size(); ++i) {
        KX_HIP(hipMemcpyAsync(bias + r0, wt(names[i] + ".bias"), (size_t)src.rows[i] * 4, hipMemcpyDeviceToDevice,
                              stream_));
        r0 += src.rows[i];
    }
    c.bias = bias;
    return c;
}

ConvW Model::make_convT(const std::string& name, int stride) {
    const TensorInfo& ti = info(name + ".weight");  // [Cin][Cout][k]
    KX_REQUIRE(ti.dims[2] == 2 * stride, "transposed conv: only k == 2*stride is supported");
    ConvW c;
    c.Cin = ti.dims[0];
    c.up_cout = ti.dims[1];
    c.up_s = stride;
    c.rows = stride * c.up_cout;
    c.K = 2;
    c.BM = conv_pick_bm(c.rows);
    c.n_chunks = (c.Cin + CONV_CK - 1) / CONV_CK;
    float* p = dev_alloc(packed_conv_floats(c.rows, c.Cin, 2, c.BM));
    launch_pack_convT(wt(name + ".weight"), p, c.Cin, c.up_cout, stride, c.BM, stream_);
    c.w = p;
    c.bias = wt(name + ".bias");
    int nf[3] = {c.Cin * c.up_cout * 2 * stride, 0, 0};
    pack16(c, PackSrc{{nullptr, nullptr, nullptr}, {0, 0, 0}}, wt(name + ".weight"), nf);
    return c;
}

LstmW Model::make_lstm(const std::string& name) {
    LstmW l;
    const TensorInfo& ti = info(name + ".weight_ih_l0");
    ConvW& c = l.ih;
    c.rows = 2048;
    c.Cin = ti.dims[1];
    c.K = 1;
    c.BM = 128;
    c.n_chunks = (c.Cin + CONV_CK - 1) / CONV_CK;
    float* p = dev_alloc(packed_conv_floats(2048, c.Cin, 1, 128));
    PackSrc src{{wt(name + ".weight_ih_l0"), wt(name + ".weight_ih_l0_reverse"), nullptr}, {1024, 1024, 0}};
    launch_pack_conv(src, p, 2048, c.Cin, 1, 128, stream_);
    c.w = p;
    int nf[3] = {1024 * c.Cin, 1024 * c.Cin, 0};
    pack16(c, src, nullptr, nf);
    float* bias = dev_alloc(2048);
    launch_vec_add(wt(name + ".bias_ih_l0"), wt(name + ".bias_hh_l0"), bias, 1024, stream_);
    launch_vec_add(wt(name + ".bias_ih_l0_reverse"), wt(name + ".bias_hh_l0_reverse"), bias + 1024, 1024, stream_);
    c.bias = bias;
    float* wh = dev_alloc(2 * 256 * 1024);
    launch_transpose_whh(wt(name + ".weight_hh_l0"), wh, stream_);
    launch_transpose_whh(wt(name + ".weight_hh_l0_reverse"), wh + 256 * 1024, stream_);
    l.whhT = wh;
    return l;
}

void Model::add_fc(const std::string& key, const std::string& fc_name, int style_off) {
    const TensorInfo& ti = info(fc_name + ".weight");
    KX_REQUIRE(ti.dims[1] == 128, "style fc: expected 128 inputs");
    FcDesc d;
    d.w = wt(fc_name + ".weight");
    d.b = wt(fc_name + ".bias");
    d.n_out = ti.dims[0];
    d.style_off = style_off;
    d.out_off = gb_total_;
    fc_off_[key] = gb_total_;
    gb_total_ += d.n_out;
    fc_host_.push_back(d);
}
long Model::fc_off(const std::string& key) const {
    auto it = fc_off_.find(key);
    if (it == fc_off_.end()) throw Error(4, "internal: unknown style fc " + key);
    return it->second;
}

void Model::build() {
    {
        const TensorInfo& we = info("bert.embeddings.word_embeddings.weight");
        const TensorInfo& te = info("text_encoder.embedding.weight");
        n_vocab_ = we.dims[0] < te.dims[0] ? we.dims[0] : te.dims[0];  // ids index both tables (178 rows each)
        KX_REQUIRE(n_vocab_ >= 1, "weight blob: empty embedding table");
    }
    const std::string L = "bert.encoder.albert_layer_groups.0.albert_layers.0.";
    convs_["bert.map"] = make_conv("bert.encoder.embedding_hidden_mapping_in");
    convs_["bert.qkv"] = make_conv_cat({L + "attention.query", L + "attention.key", L + "attention.value"});
    convs_["bert.dense"] = make_conv(L + "attention.dense");
    convs_["bert.ffn"] = make_conv(L + "ffn");
    convs_["bert.ffn_out"] = make_conv(L + "ffn_output");
    convs_["bert_encoder"] = make_conv("bert_encoder");
    for (int i = 0; i < 3; ++i) {
        const std::string n = "predictor.text_encoder.lstms." + std::to_string(2 * i);
        lstms_[n] = make_lstm(n);
        add_fc("dur_enc." + std::to_string(i), "predictor.text_encoder.lstms." + std::to_string(2 * i + 1) + ".fc", 128);
    }
    lstms_["predictor.lstm"] = make_lstm("predictor.lstm");
    lstms_["predictor.shared"] = make_lstm("predictor.shared");
    lstms_["text_encoder.lstm"] = make_lstm("text_encoder.lstm");
    convs_["duration_proj"] = make_conv("predictor.duration_proj.linear_layer");
    auto reg_resblk = [&](const std::string& n, int style_off) {
        convs_[n + ".conv1"] = make_conv(n + ".conv1");
        convs_[n + ".conv2"] = make_conv(n + ".conv2");
        if (has(n + ".conv1x1.weight")) convs_[n + ".conv1x1"] = make_conv(n + ".conv1x1", false);
        add_fc(n + ".norm1", n + ".norm1.fc", style_off);
        add_fc(n + ".norm2", n + ".norm2.fc", style_off);
    };
    for (const char* br : {"F0", "N"}) {
        for (int i = 0; i < 3; ++i) reg_resblk(std::string("predictor.") + br + "." + std::to_string(i), 128);
        convs_[std::string("predictor.") + br + "_proj"] = make_conv(std::string("predictor.") + br + "_proj");
    }
    for (int i = 0; i < 3; ++i)
        convs_["text_encoder.cnn." + std::to_string(i)] = make_conv("text_encoder.cnn." + std::to_string(i) + ".0");
    reg_resblk("decoder.encode", 0);
    for (int i = 0; i < 4; ++i) reg_resblk("decoder.decode." + std::to_string(i), 0);
    convs_["decoder.F0_conv"] = make_conv("decoder.F0_conv");
    convs_["decoder.N_conv"] = make_conv("decoder.N_conv");
    convs_["decoder.asr_res"] = make_conv("decoder.asr_res.0");
    const std::string G = "decoder.generator.";
    auto reg_resblock1 = [&](const std::string& n) {
        for (int i = 0; i < 3; ++i) {
            const std::string s = std::to_string(i);
            convs_[n + ".convs1." + s] = make_conv(n + ".convs1." + s);
            convs_[n + ".convs2." + s] = make_conv(n + ".convs2." + s);
            add_fc(n + ".adain1." + s, n + ".adain1." + s + ".fc", 0);
            add_fc(n + ".adain2." + s, n + ".adain2." + s + ".fc", 0);
        }
    };
    for (int i = 0; i < 2; ++i) {
        convs_[G + "noise_convs." + std::to_string(i)] = make_conv(G + "noise_convs." + std::to_string(i));
        reg_resblock1(G + "noise_res." + std::to_string(i));
    }
    convs_[G + "ups.0"] = make_convT(G + "ups.0", 10);
    convs_[G + "ups.1"] = make_convT(G + "ups.1", 6);
    for (int i = 0; i < 6; ++i) reg_resblock1(G + "resblocks." + std::to_string(i));
    convs_[G + "conv_post"] = make_conv(G + "conv_post");

    for (auto& kv : convs_) kv.second.name = kv.first;
    if (conv_mode == CONV_BF16 || conv_mode == CONV_F16F8) set_conv_mode(conv_mode);  // (KOKOROX_CONV=bf16 / f16f8: the images exist from the start)
    for (auto& kv : lstms_) kv.second.ih.name = kv.first + ".ih";
    KX_HIP(hipMalloc((void**)&fc_dev_, fc_host_.size() * sizeof(FcDesc)));
    owned_.push_back(fc_dev_);
    KX_HIP(hipMemcpyAsync(fc_dev_, fc_host_.data(), fc_host_.size() * sizeof(FcDesc), hipMemcpyHostToDevice, stream_));
    KX_HIP(hipStreamSynchronize(stream_));
}

// ---- helpers ------------------------------------------------------------------------------------
void Model::ensure_arena(Arena& a, size_t bytes) {
    if (bytes <= a.cap) return;
    KX_HIP(hipStreamSynchronize(stream_));
    if (a.base) KX_HIP(hipFree(a.base));
    a.base = nullptr;
    a.cap = 0;
    // half again as much as asked for: a serving process meets its largest (batch x length) shape step by step, and every
    // regrowth is a stream sync + hipFree + hipMalloc of gigabytes (seen as a 1.3 s latency outlier in a 25 s soak with 12 %
    // slack); the card has 288 GB
    const size_t want = bytes + bytes / 2 + (1 << 20);
    KX_HIP(hipMalloc((void**)&a.base, want));
    a.cap = want;
}



// ---- lanes (model.h) ---------------------------------------------------------------------------
hipEvent_t Model::record_here() {
    if (dry_) return nullptr;
    if (lane_ev_used_ == lane_ev_.size()) {
        hipEvent_t e;
        KX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        lane_ev_.push_back(e);
    }
    hipEvent_t e = lane_ev_[lane_ev_used_++];
    KX_HIP(hipEventRecord(e, stream_));
    return e;
}

void Model::wait_here(hipEvent_t e) {
    if (dry_ || !e) return;
    KX_HIP(hipStreamWaitEvent(stream_, e, 0));
}

void Model::sync_lanes() {
    for (int i = 1; i < N_LANES; ++i)
        if (lanes_[i].stream) (void)hipStreamSynchronize(lanes_[i].stream);
}

// Everything issued inside the scope goes to lane k (its stream, its InstanceNorm parameter set), which first waits for
// what the issuing stream has queued so far.  Leaving the scope joins nothing: chains meet through record_here / wait_here.
struct Model::LaneScope {
    Model& m;
    hipStream_t s0;
    float *a0, *b0, *c0;
    LaneScope(Model& mm, int k) : m(mm), s0(mm.stream_), a0(mm.nmean_), b0(mm.nscale_), c0(mm.nshift_) {
        const Lane& L = m.lanes_[k < m.n_lanes_ ? k : 0];
        if (L.stream == m.stream_) return;
        hipEvent_t e = m.record_here();
        m.stream_ = L.stream;
        m.nmean_ = L.nmean;
        m.nscale_ = L.nscale;
        m.nshift_ = L.nshift;
        m.wait_here(e);
    }
    ~LaneScope() {
        m.stream_ = s0;
        m.nmean_ = a0;
        m.nscale_ = b0;
        m.nshift_ = c0;
    }
};

void Model::conv(const ConvW& w, const T& in, const T& out, const ConvOpts& o) {
    // Layers whose input window many row tiles stage get a pre-split image of the input (conv_f16x3_pre.hip): planned in the
    // dry run like every other buffer of the back half; the rule looks at the layer's shape only.
    void* x16 = nullptr;
    long x16_bs = 0;
    if ((conv_mode == CONV_F16X3 || conv_mode == CONV_F16F8) && img_arena_ && conv16_pre_shape(w.BM, w.rows, w.K, o.dil, o.stride, o.act, o.in_up2)) {
        x16_bs = (long)conv16_pre_image_bytes(w.Cin, in.ld);
        x16 = img_arena_->alloc((size_t)B_ * x16_bs);
    }
    if (dry_) return;
    ConvArgs a{};
    a.x = in.p;
    a.x_bs = in.bs;
    a.x_ld = in.ld;
    a.Cin = w.Cin;
    KX_REQUIRE(in.C == w.Cin, "internal: conv Cin mismatch");
    a.in_len = in.len;
    if (o.in_up2) {
        a.in_len.mul *= 2;
        a.in_len.add *= 2;
    }
    a.out_len = (o.store == ST_UPSCATTER) ? o.up_len : out.len;
    a.w = w.w;
    a.bias = w.bias;
    a.nmean = o.nmean;
    a.nscale = o.nscale;
    a.nshift = o.nshift;
    a.n_bs = n_bs_;
    a.act = o.act;
    a.slope = o.slope;
    a.alpha = o.alpha;
    a.K = w.K;
    a.dil = o.dil;
    a.stride = o.stride;
    a.pad = o.pad;
    a.in_up2 = o.in_up2;
    a.Cout = w.rows;
    a.n_chunks = w.n_chunks;
    a.y = out.p;
    a.y_bs = out.bs;
    a.y_ld = out.ld;
    if (o.resid) {
        a.resid = o.resid->p;
        a.r_bs = o.resid->bs;
        a.r_ld = o.resid->ld;
    }
    a.accum = o.accum;
    a.out_mul = o.out_mul;
    a.out_div = o.out_div;
    a.epi = o.epi;
    a.store = o.store;
    a.up_s = w.up_s;
    a.up_pad = o.up_pad;
    a.up_off = o.up_off;
    a.up_reflect = o.up_reflect;
    a.up_cout = w.up_cout ? w.up_cout : 1;
    const bool f16 = conv_mode == CONV_F16X3 || conv_mode == CONV_F16 || conv_mode == CONV_BF16 || conv_mode == CONV_F16F8;
    // reduced-precision mode (opt-in): the decoder and generator convs that take the direct-A kernel run one f16 MFMA
    // per product; everything upstream of the F0 / N curves (duration head, prosody predictor) and every kernel that is
    // not the direct-A conv (harmonic source, STFT pair, k = 1 GEMMs, conv_post) stays f32-class (hard part 3)
    a.prec1 = (conv_mode == CONV_F16 && p1_region_) ? 1 : ((conv_mode == CONV_BF16 && p1_region_ && w.w16b) ? 2 : 0);
    a.w16 = w.w16;
    a.w16b = w.w16b;
    // f16f8 mode (opt-in): the layers that carry an 8-bit cross image run two MFMA-equivalents per product instead of three
    a.w8x = conv_mode == CONV_F16F8 ? w.w8x : nullptr;
    // outputs that no cache can hold until the next layer reads them (> 512 MB: L2 is 32 MB, MALL 256 MB) are streamed by the direct-A
    // kernels' interior stores (non-temporal stores and residual loads); smaller ones (small batches, the token axis, the decoder)
    // stay cacheable
    static const long stream_mb = getenv("KX_EPI_STREAM_MB") ? atol(getenv("KX_EPI_STREAM_MB")) : 512;
    a.epi_stream = f16 && stream_mb >= 0 && (double)B_ * w.rows * out.ld * 4.0 > (double)stream_mb * 1048576.0;
    a.n_chunks16 = w.n_chunks16;
    static const int xcd_swz = getenv("KX_XCD_SWIZZLE") ? atoi(getenv("KX_XCD_SWIZZLE")) : 1;
    a.xcd_swizzle = xcd_swz;
    a.x_prescale = std::ldexp(1.0f, w.act_shift);
    a.w_unscale = std::ldexp(w.unscale, -w.act_shift);  // (exact: both are powers of two)
    if (diag_on_ && diag_used_ < diag_cap_) {
        float* slot = d_diag_ + 3 * diag_used_++;
        launch_diag_stats(in.p, in.bs, in.ld, w.Cin, in.len, B_, in.Lmax, o.nmean, o.nscale, o.nshift, n_bs_, slot, stream_);
        diag_recs_.push_back(DiagRec{w.name, w.rows, w.Cin, w.K, w.act_shift, 0.0, 0.0, 0.0});
    }
    parts_.erase(out.p);  // whatever statistics were known for this tensor are stale now
    static const bool fuse_stats = !(getenv("KX_FUSE_STATS") && atoi(getenv("KX_FUSE_STATS")) == 0);
    if (fuse_stats && o.stat_part && o.store == ST_NORMAL && !o.accum) {
        const int max_c = out.Lmax;
        int bn, wn;
        if (f16) {
            conv16_pick_tile(w.BM, max_c, B_, w.rows, w.K, o.dil, o.stride, &bn, &wn, 0, true, o.act, w.n_chunks16, conv16_pmode(a));
        } else {
            bn = conv_bn(w.BM);
            wn = w.BM == 128 ? 2 : 4;
        }
        a.stat_part = o.stat_part;
        a.stat_tiles = ((max_c + bn - 1) / bn) * wn;
        parts_[out.p] = PartInfo{o.stat_part, a.stat_tiles, bn / wn, w.rows};
    }
    static const int dbg_env = getenv("KX_DBG") ? atoi(getenv("KX_DBG")) : 0;
    a.dbg = dbg_env;
    int max_cols = (o.store == ST_UPSCATTER) ? in.Lmax + 1 : out.Lmax;
    static const bool merge_env = !(getenv("KX_MERGE") && atoi(getenv("KX_MERGE")) == 0);
    if (merge_env && f16 && B_ > 1 && w.K == 1 && o.stride == 1 && o.pad == 0 && !o.in_up2 && !o.nmean &&
        o.store != ST_UPSCATTER && !o.stat_part && in.Lmax <= 512 && in.len.mul == 1 && in.len.add == 0 &&
        out.len.lens == in.len.lens && out.len.mul == 1 && out.len.add == 0) {
        a.merge_T = in.Lmax;  // k = 1 GEMM on a short axis: one merged column space for the whole batch
        a.merge_B = B_;
        max_cols = B_ * in.Lmax;
    }
    // ragged batch: the direct-A kernels take a flat list of the live tiles instead of a (longest length) x B grid
    if (f16 && B_ > 1) {
        const int fbn = conv16_flat_bn(a, w.BM, B_, max_cols);
        if (fbn) {
            const LenMap& lm = (o.store == ST_UPSCATTER) ? a.in_len : a.out_len;
            const int extra = o.store == ST_UPSCATTER ? 1 : 0;
            int total = 0;
            a.tile_prefix = tile_prefix_for(lm, extra, fbn, &total);
            a.flat_ny = (w.rows + 127) / 128;
            a.flat_B = B_;
            a.flat_tiles_host = total;
            a.flat_bn_host = fbn;
            if (total <= 0) return;  // (nothing to compute)
        }
    }
    // diagnostic: KX_STAMP=<file> dumps per-workgroup timestamps of the first 128->128 k=11 launch
    static const char* stamp_path = getenv("KX_STAMP");
    static bool stamped = false;
    unsigned long long* d_stamps = nullptr;
    long n_wg = 0;
    static const int stamp_rows = getenv("KX_STAMP_ROWS") ? atoi(getenv("KX_STAMP_ROWS")) : 128;
    static const int stamp_k = getenv("KX_STAMP_K") ? atoi(getenv("KX_STAMP_K")) : 11;
    static int stamp_skip = getenv("KX_STAMP_SKIP") ? atoi(getenv("KX_STAMP_SKIP")) : 0;  // matching launches to pass over first
    const bool stamp_match = stamp_path && !stamped && f16 && w.K == stamp_k && w.rows == stamp_rows && B_ >= 8 && !dry_;
    if (stamp_match && stamp_skip > 0) --stamp_skip;
    else if (stamp_match) {
        n_wg = (long)((max_cols + 127) / 128) * ((w.rows + 127) / 128) * (a.merge_T > 0 ? 1 : B_);
        KX_HIP(hipMalloc((void**)&d_stamps, n_wg * 64));
        KX_HIP(hipMemsetAsync(d_stamps, 0, n_wg * 64, stream_));
        a.stamps = d_stamps;
    }
    if (x16_bs) {
        // (outside the timed interval of the profile mode below: that one is the conv kernel's own duration, which the rocprofv3
        // summary of the same kernel name must reproduce; the pass shows up under its own name there and in ms_per_step)
        launch_split_image(a, B_, in.Lmax, x16, x16_bs, stream_);
        a.x16 = x16;
        a.x16_bs = x16_bs;
        a.x16_ld = in.ld;
    }
    if (prof_on_ && w.BM == 128) {  // the dominant kernel family: every 128-row conv / GEMM launch (direct-A, direct-A GEMM, LDS-DMA forms; f32 mode: conv1d_mfma_kernel<128,128,2,2>)
        const LenMap& lm = (o.store == ST_UPSCATTER) ? in.len : out.len;
        const std::vector<int>& hl = (lm.lens == dT_) ? hT_ : hF_;
        double cols = 0;
        for (int b = 0; b < B_; ++b) cols += (double)hl[b] * lm.mul + lm.add + (o.store == ST_UPSCATTER ? 1 : 0);
        prof_flops_ += 2.0 * w.rows * w.Cin * w.K * cols;
        prof_launches_ += 1;
        // algorithmic HBM bytes of the launch: the input tensor once, the residual and the running
        // sum where the epilogue reads them, the output once, the split-f16 weights once
        double in_cols = 0;
        {
            const std::vector<int>& hi = (in.len.lens == dT_) ? hT_ : hF_;
            for (int b = 0; b < B_; ++b) in_cols += (double)hi[b] * in.len.mul + in.len.add;
        }
        const double out_rows = (o.store == ST_UPSCATTER) ? (double)(w.up_cout ? w.up_cout : 1) : (double)w.rows;
        const double out_cols = (o.store == ST_UPSCATTER) ? cols * w.up_s : cols;
        const double out_elems = out_rows * out_cols;
        const double bytes = 4.0 * ((double)w.Cin * in_cols + out_elems * (1.0 + (o.resid ? 1.0 : 0.0) + (o.accum ? 1.0 : 0.0)) +
                                    (double)w.rows * w.Cin * w.K);
        prof_recs_.push_back(ProfRec{w.rows, w.Cin, w.K, o.dil, o.stride, o.store, cols, 2.0 * w.rows * w.Cin * w.K * cols, 0.f, bytes});
        if (ev_used_ + 2 > ev_.size()) {
            for (int i = 0; i < 64; ++i) {
                hipEvent_t e;
                KX_HIP(hipEventCreate(&e));
                ev_.push_back(e);
            }
        }
        KX_HIP(hipEventRecord(ev_[ev_used_], stream_));
        if (f16) launch_conv1d_f16x3(a, w.BM, B_, max_cols, stream_);
        else launch_conv1d(a, w.BM, B_, max_cols, stream_);
        KX_HIP(hipEventRecord(ev_[ev_used_ + 1], stream_));
        ev_used_ += 2;
    } else {
        if (f16) launch_conv1d_f16x3(a, w.BM, B_, max_cols, stream_);
        else launch_conv1d(a, w.BM, B_, max_cols, stream_);
    }
    if (d_stamps) {
        stamped = true;
        KX_HIP(hipStreamSynchronize(stream_));
        std::vector<unsigned long long> hst(n_wg * 8);
        KX_HIP(hipMemcpy(hst.data(), d_stamps, n_wg * 64, hipMemcpyDeviceToHost));
        KX_HIP(hipFree(d_stamps));
        if (FILE* f = fopen(stamp_path, "wb")) {
            fwrite(hst.data(), 8, hst.size(), f);
            fclose(f);
        }
    }
}

// The device prefix table of (length map, extra columns, tile width) for the running call: built once per call and key by a
// one-thread kernel on the current stream, from the same device lengths the kernels read; *total = its last entry, counted
// on the host from the host copies of those lengths (the grid size).
const int* Model::tile_prefix_for(const LenMap& lm, int extra, int bn, int* total) {
    const std::vector<int>& hl = (lm.lens == dT_) ? hT_ : hF_;
    KX_REQUIRE(lm.lens == dT_ || lm.lens == dF_, "internal: tile prefix of an unknown length array");
    int tot = 0;
    for (int b = 0; b < B_; ++b) {
        const int cols = hl[b] * lm.mul + lm.add + extra;
        tot += cols > 0 ? (cols + bn - 1) / bn : 0;
    }
    *total = tot;
    for (const PrefixKey& k : prefix_keys_)
        if (k.lens == lm.lens && k.mul == lm.mul && k.add == lm.add + extra && k.bn == bn && k.stream == stream_) return k.dev;
    const size_t need = (size_t)(B_ + 1);
    if (prefix_used_ + need > prefix_cap_) {  // (grown like the arenas; tables of this call that are in use stay where they are)
        const size_t want = std::max<size_t>(prefix_cap_ * 2, (size_t)64 * need);
        int* p = nullptr;
        KX_HIP(hipMalloc((void**)&p, want * sizeof(int)));
        owned_.push_back(p);  // (the old block stays alive until the model goes: launches of this call may still read it)
        d_prefix_ = p;
        prefix_cap_ = want;
        prefix_used_ = 0;
    }
    int* dev = d_prefix_ + prefix_used_;
    prefix_used_ += need;
    launch_tile_prefix(lm, extra, bn, B_, dev, stream_);
    // (keyed by stream too: a table built on one lane's stream is ordered before that lane's launches only)
    prefix_keys_.push_back(PrefixKey{lm.lens, lm.mul, lm.add + extra, bn, stream_, dev});
    return dev;
}

void Model::stats(const T& x, const std::string& fc_key) {
    // a tensor whose sums are not known yet gets a small cache for them (planned in the dry run like everything else)
    float2* raw = stats_arena_ ? reinterpret_cast<float2*>(stats_arena_->alloc((size_t)B_ * x.C * 2 * sizeof(float2))) : nullptr;
    if (dry_) return;
    auto it = parts_.find(x.p);
    if (it != parts_.end() && it->second.C == x.C) {
        const PartInfo& pi = it->second;
        launch_stats_finalize(pi.part, pi.tiles, pi.cols_per_tile, x.C, x.len, B_, gb_ + fc_off(fc_key), gb_total_,
                              nmean_, nscale_, nshift_, n_bs_, stream_);
        return;
    }
    if (prof_on_) {  // (the PMC tooling checks FETCH_SIZE of this kernel against these bytes: it reads x exactly once)
        const std::vector<int>& hl = (x.len.lens == dT_) ? hT_ : hF_;
        double cols = 0;
        for (int b = 0; b < B_; ++b) cols += (double)hl[b] * x.len.mul + x.len.add;
        prof_stats_bytes_ += 4.0 * x.C * cols;
        prof_stats_launches_ += 1;
    }
    launch_in_stats(x.p, x.bs, x.ld, x.C, x.len, B_, gb_ + fc_off(fc_key), gb_total_, nmean_, nscale_, nshift_, n_bs_,
                    raw, stream_);
    // (two "tiles" of one column each: the high and the low part of the f64 sums)
    if (raw) parts_[x.p] = PartInfo{raw, 2, 1, x.C};
}

void Model::tap(const char* name, const T& t) {
    if (!taps_on_ || dry_) return;
    KX_HIP(hipStreamSynchronize(stream_));
    Tap tp;
    tp.B = B_;
    tp.C = t.C;
    tp.ld = t.ld;
    tp.data.resize((size_t)B_ * t.C * t.ld);
    const std::vector<int>& hl = (t.len.lens == dT_) ? hT_ : hF_;
    for (int b = 0; b < B_; ++b) {
        tp.L.push_back(hl[b] * t.len.mul + t.len.add);
        KX_HIP(hipMemcpy(tp.data.data() + (size_t)b * t.C * t.ld, t.p + (long)b * t.bs, (size_t)t.C * t.ld * 4,
                         hipMemcpyDeviceToHost));
    }
    taps_[name] = std::move(tp);
}

const Tap* Model::find_tap(const std::string& name) const {
    auto it = taps_.find(name);
    return it == taps_.end() ? nullptr : &it->second;
}

void Model::lstm(const LstmW& w, const T& in, const T& out, float* gx) {
    if (dry_) return;
    T g;
    g.p = gx;
    g.bs = (long)in.Lmax * 2048;
    g.ld = 2048;
    g.C = 2048;
    g.len = in.len;
    g.Lmax = in.Lmax;
    ConvOpts o;
    o.store = ST_TMAJOR;
    conv(w.ih, in, g, o);
    // (after a timed-out hand-off the model stays on the one-CU kernel: see check_dev_err)
    const int xb = stream_ == main_stream_ ? 0 : 1;
    launch_lstm(gx, g.bs, 2048, w.whhT, out.p, out.bs, out.ld, in.len, B_, lstm_pair_ok_ ? d_xchg_[xb] : nullptr,
                d_dev_err_, stream_, &xchg_epoch_[xb]);
}

// AdainResBlk1d (istftnet.py): out = (conv2(act(norm2(conv1(pool(act(norm1(x))))))) + shortcut(x)) / sqrt(2)
void Model::adain_resblk(const std::string& name, const T& x, const T& out, bool upsample, float* ws_a, float* ws_b,
                         float* ws_c) {
    const ConvW& c1 = convs_.at(name + ".conv1");
    const ConvW& c2 = convs_.at(name + ".conv2");
    T t1 = out;
    t1.p = ws_a;
    t1.bs = (long)out.C * out.ld;
    // the 1x1 shortcut depends on x only: it is issued first, on a side lane, and joins before conv2 reads it
    T sc = out;
    const T* res = &x;
    hipEvent_t ev_sc = nullptr;
    if (convs_.count(name + ".conv1x1")) {
        sc.p = ws_b;
        sc.bs = (long)out.C * out.ld;
        ConvOpts osc;
        osc.in_up2 = upsample ? 1 : 0;
        {
            LaneScope side(*this, 2);
            conv(convs_.at(name + ".conv1x1"), x, sc, osc);
            ev_sc = record_here();
        }
        res = &sc;
    } else {
        KX_REQUIRE(!upsample && x.C == out.C, "internal: identity shortcut needs equal shapes");
    }
    // InstanceNorm partial sums of t1 (normalised by norm2 below) and of the block's output (normalised by the next block's
    // norm1 when that block reads exactly this tensor) leave the conv epilogues, as in the generator: no separate pass
    auto part_for = [&](const T& t) -> float2* {
        const size_t n = (size_t)B_ * t.C * (t.Lmax / 64 + 4);
        return stats_arena_ ? static_cast<float2*>(stats_arena_->alloc(n * sizeof(float2))) : nullptr;
    };
    float2* part_t1 = part_for(t1);
    float2* part_out = part_for(out);
    stats(x, name + ".norm1");
    if (!upsample) {
        ConvOpts o;
        o.nmean = nmean_; o.nscale = nscale_; o.nshift = nshift_;
        o.act = ACT_LEAKY; o.slope = 0.2f; o.pad = 1;
        o.stat_part = part_t1;
        conv(c1, x, t1, o);
    } else {
        T p = out;
        p.p = ws_c;
        p.C = x.C;
        p.bs = (long)x.C * out.ld;
        if (!dry_)
            launch_pool_up2(x.p, x.bs, x.ld, x.C, nmean_, nscale_, nshift_, n_bs_, 0.2f, wt(name + ".pool.weight"),
                            wt(name + ".pool.bias"), p.p, p.bs, p.ld, x.len, B_, x.Lmax, stream_);
        ConvOpts o;
        o.pad = 1;
        o.stat_part = part_t1;
        conv(c1, p, t1, o);
    }
    stats(t1, name + ".norm2");
    ConvOpts o;
    o.nmean = nmean_; o.nscale = nscale_; o.nshift = nshift_;
    o.act = ACT_LEAKY; o.slope = 0.2f; o.pad = 1;
    o.resid = res;
    o.out_mul = RSQRT2;
    o.stat_part = part_out;
    wait_here(ev_sc);  // (the shortcut ran beside conv1)
    conv(c2, t1, out, o);
}

// AdaINResBlock1 with Snake1D (istftnet.py).  x is read-only; xj/t1 are scratch of x's shape;
// the third iteration lands in `out` (optionally accumulated and divided: mean over kernels).
void Model::adain_resblock1(const std::string& name, int k, const T& x, const T& xj, const T& t1, const T& out,
                            int accum, float out_div, float2* part_t1, float2* part_xj, hipEvent_t wait_before_last) {
    static const int dils[3] = {1, 3, 5};
    for (int i = 0; i < 3; ++i) {
        const std::string s = std::to_string(i);
        const T& cur = (i == 0) ? x : xj;
        const T& dst = (i == 2) ? out : xj;
        stats(cur, name + ".adain1." + s);
        ConvOpts o1;
        o1.nmean = nmean_; o1.nscale = nscale_; o1.nshift = nshift_;
        o1.act = ACT_SNAKE;
        o1.alpha = dry_ ? nullptr : wt(name + ".alpha1." + s);
        o1.dil = dils[i];
        o1.pad = (k * dils[i] - dils[i]) / 2;
        o1.stat_part = part_t1;  // t1 is normalised by adain2 next
        conv(convs_.at(name + ".convs1." + s), cur, t1, o1);
        stats(t1, name + ".adain2." + s);
        ConvOpts o2;
        o2.nmean = nmean_; o2.nscale = nscale_; o2.nshift = nshift_;
        o2.act = ACT_SNAKE;
        o2.alpha = dry_ ? nullptr : wt(name + ".alpha2." + s);
        o2.pad = (k - 1) / 2;
        o2.resid = &cur;
        if (i == 2) {
            o2.accum = accum;
            o2.out_div = out_div;
            wait_here(wait_before_last);  // (the running sum this conv adds to is written by another lane)
        } else {
            o2.stat_part = part_xj;  // xj is normalised by the next iteration's adain1
        }
        conv(convs_.at(name + ".convs2." + s), t1, dst, o2);
    }
}

void Model::sync() {
    KX_HIP(hipSetDevice(device));
    KX_HIP(hipStreamSynchronize(stream_));
    check_dev_err();
}

// (the stream is idle) raise what a kernel recorded in the sticky device error word, and clear it
void Model::check_dev_err() {
    // (through the model's own stream into page-locked memory: a null-stream copy here would wait for, and hold up, every
    // other model's blocking work on this GPU)
    KX_HIP(hipMemcpyAsync(&h_words_[0], d_dev_err_, sizeof(unsigned), hipMemcpyDeviceToHost, main_stream_));
    KX_HIP(hipStreamSynchronize(main_stream_));
    const unsigned e = h_words_[0];
    if (!e) return;
    KX_HIP(hipMemsetAsync(d_dev_err_, 0, sizeof(unsigned), main_stream_));
    KX_HIP(hipStreamSynchronize(main_stream_));
    // A part of a resident-weights recurrence never saw its partner (starved behind other work on the GPU): what this call
    // computed is garbage.  The model switches to the streaming recurrence -- same bits, no partner to wait for -- and goes back
    // to the resident forms after LSTM_REARM_AFTER clean forwards; the host entry points re-run the call once (infer_host_ex).
    lstm_pair_ok_ = false;
    lstm_rearm_in_ = LSTM_REARM_AFTER;
    n_lstm_timeouts_ += 1;
    throw LstmTimeout("device error word " + std::to_string(e) +
                      ": a part of the resident-weights LSTM recurrence never saw its partner; this call's result is invalid; the "
                      "model runs the streaming recurrence (same bits) for the next " + std::to_string(LSTM_REARM_AFTER) + " forwards");
}

void Model::set_conv_mode(int mode) {
    sync();
    if (mode == CONV_BF16) {
        // bf16 forms of the direct-A kernels' weight images, once: bf16(hi + lo) from the split-f16 image (160 MB more)
        KX_HIP(hipSetDevice(device));
        for (auto& kv : convs_) {
            ConvW& c = kv.second;
            if (c.w16b || !c.w16 || c.BM != 128) continue;
            const size_t nh = packed_conv16_halves(c.rows, c.Cin, c.K, c.BM);
            void* p = nullptr;
            KX_HIP(hipMalloc(&p, nh * 2));
            owned_.push_back(p);
            launch_image_to_bf16(c.w16, p, nh, stream_);
            c.w16b = p;
        }
        KX_HIP(hipStreamSynchronize(stream_));
    }
    if (mode == CONV_F16F8) {
        // 8-bit cross images of the layers the f16f8 kernels take (7- and 11-tap convs, and the 3-tap ones of at most 256 rows: the
        // generator's snake resblocks -- the activation is a property of the call, not of the weights, so a few leaky 3-tap convs of
        // the predictor get an image they never use), once
        KX_HIP(hipSetDevice(device));
        for (auto& kv : convs_) {
            ConvW& c = kv.second;
            const bool taps = c.K == 7 || c.K == 11 || (c.K == 3 && c.rows <= 256 && conv16_da_f8_shape(3, 1));
            if (c.w8x || !c.w16 || c.BM != 128 || c.up_s || !taps || c.n_chunks16 < 2 || (c.n_chunks16 & 1)) continue;
            void* p = nullptr;
            KX_HIP(hipMalloc(&p, packed_conv8x_bytes(c.rows, c.Cin, c.K)));
            owned_.push_back(p);
            launch_pack_conv8x(c.w16, p, c.rows, c.Cin, c.K, stream_);
            c.w8x = p;
        }
        KX_HIP(hipStreamSynchronize(stream_));
    }
    conv_mode = mode;
}

void Model::info(int64_t out[8]) const {
    out[0] = source_variant_;
    out[1] = (source_variant_ == 3 || source_variant_ == 4) ? 0 : 1;
    out[2] = conv_mode;
    out[3] = n_vocab_;
    out[4] = n_voices_.load(std::memory_order_acquire);
    out[5] = part_;
    out[6] = n_parts_;
    int cus = cu_count_;
    if (cus == 0) {
        hipDeviceProp_t prop;
        cus = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 0;
    }
    out[7] = cus;
}

void Model::status(int64_t out[4]) const {
    out[0] = lstm_pair_ok_ ? 0 : 1;
    out[1] = n_lstm_timeouts_;
    out[2] = lstm_pair_ok_ ? 0 : lstm_rearm_in_;
    out[3] = n_rerun_;
}

// a forward has finished cleanly: count down to the resident-weights recurrence's return
void Model::note_clean_forward() {
    if (!lstm_pair_ok_ && lstm_rearm_in_ > 0 && --lstm_rearm_in_ == 0) lstm_pair_ok_ = true;
}

void Model::set_pinned(const int32_t* pattern, int n) {
    KX_HIP(hipSetDevice(device));
    KX_HIP(hipStreamSynchronize(stream_));
    n_pinned_ = 0;
    if (n <= 0) return;
    KX_REQUIRE(pattern && n <= 512, "pinned durations: 1..512 entries");
    for (int i = 0; i < n; ++i) KX_REQUIRE(pattern[i] >= 1 && pattern[i] <= 50, "pinned durations must be in 1..50");
    if (!d_pinned_) {
        KX_HIP(hipMalloc((void**)&d_pinned_, 512 * sizeof(int)));
        owned_.push_back(d_pinned_);
    }
    KX_HIP(hipMemcpy(d_pinned_, pattern, n * sizeof(int), hipMemcpyHostToDevice));
    n_pinned_ = n;
}

void Model::warmup(int B, int n_tokens, int frames_per_token) {
    KX_REQUIRE(B >= 1 && B <= 4096 && n_tokens >= 2 && n_tokens <= 512 && frames_per_token >= 1 && frames_per_token <= 50,
               "warmup: 1..4096 utterances of 2..512 tokens at 1..50 frames per token");
    // (the caller's pinned pattern, if any, is put back afterwards)
    std::vector<int32_t> saved((size_t)n_pinned_);
    if (n_pinned_) {
        KX_HIP(hipSetDevice(device));
        KX_HIP(hipStreamSynchronize(stream_));
        KX_HIP(hipMemcpy(saved.data(), d_pinned_, saved.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    const int32_t fpt = frames_per_token;
    set_pinned(&fpt, 1);
    struct Restore {
        Model& m;
        std::vector<int32_t>& s;
        ~Restore() {
            try {
                m.set_pinned(s.empty() ? nullptr : s.data(), (int)s.size());
            } catch (...) {
            }
        }
    } restore{*this, saved};
    std::vector<int64_t> ids((size_t)B * n_tokens, 1);
    for (int b = 0; b < B; ++b) ids[(size_t)b * n_tokens] = ids[(size_t)b * n_tokens + n_tokens - 1] = 0;  // the two pads
    std::vector<int32_t> lens((size_t)B, n_tokens);
    std::vector<float> styles((size_t)B * 256, 0.f);
    const float speed = 1.f;
    HostCall hc;
    hc.styles = styles.data();
    void* out = nullptr;
    std::vector<int64_t> bytes((size_t)B), samples((size_t)B);
    infer_host_ex(ids.data(), n_tokens, lens.data(), B, &speed, 1, 0, 1u /* noise off */, hc, &out, bytes.data(), samples.data());
    host_out_free(out);
}

void Model::diag_enable(bool on) {
    sync();
    diag_on_ = on;
    diag_recs_.clear();
    diag_used_ = 0;
    if (on && !d_diag_) {
        diag_cap_ = 1024;
        d_diag_ = dev_alloc(3 * diag_cap_);
    }
    if (on) {
        KX_HIP(hipMemsetAsync(d_diag_, 0, 3 * diag_cap_ * sizeof(float), stream_));
        KX_HIP(hipStreamSynchronize(stream_));
    }
}

const std::vector<Model::DiagRec>& Model::diag_collect() {
    sync();
    std::vector<float> h(3 * diag_used_);
    if (diag_used_) KX_HIP(hipMemcpy(h.data(), d_diag_, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < diag_recs_.size() && i < diag_used_; ++i) {
        diag_recs_[i].absmax = h[3 * i];
        diag_recs_[i].count = (double)h[3 * i + 2];  // (one add of the utterance's length per channel and utterance)
        diag_recs_[i].rms = diag_recs_[i].count > 0 ? std::sqrt((double)h[3 * i + 1] / diag_recs_[i].count) : 0.0;
    }
    return diag_recs_;
}

void Model::set_act_shift(const std::string& conv_name, int shift) {
    KX_REQUIRE(shift >= -24 && shift <= 24, "activation pre-scale: shift must be in -24..24");
    sync();
    auto it = convs_.find(conv_name);
    if (it != convs_.end()) {
        it->second.act_shift = shift;
        return;
    }
    const std::string suf = ".ih";
    if (conv_name.size() > suf.size() && conv_name.compare(conv_name.size() - suf.size(), suf.size(), suf) == 0) {
        auto il = lstms_.find(conv_name.substr(0, conv_name.size() - suf.size()));
        if (il != lstms_.end()) {
            il->second.ih.act_shift = shift;
            return;
        }
    }
    throw Error(4, "no such conv layer: " + conv_name);
}

int Model::get_act_shift(const std::string& conv_name) const {
    auto it = convs_.find(conv_name);
    if (it == convs_.end()) throw Error(4, "no such conv layer: " + conv_name);
    return it->second.act_shift;
}

void Model::profile_enable(bool on) {
    sync();
    prof_on_ = on;
    prof_recs_.clear();
    ev_used_ = 0;
    prof_flops_ = 0;
    prof_launches_ = 0;
    prof_stats_bytes_ = 0;
    prof_stats_launches_ = 0;
}

void Model::profile_aux(int64_t* stats_launches, double* stats_bytes) {
    if (!prof_on_) throw Error(4, "profiling is not enabled");
    *stats_launches = prof_stats_launches_;
    *stats_bytes = prof_stats_bytes_;
    prof_stats_launches_ = 0;
    prof_stats_bytes_ = 0;
}

void Model::profile_read(int64_t* launches, double* ms, double* flops) {
    if (!prof_on_) throw Error(4, "profiling is not enabled");
    sync();
    double total = 0;
    for (size_t i = 0; i + 1 < ev_used_; i += 2) {
        float t = 0;
        KX_HIP(hipEventElapsedTime(&t, ev_[i], ev_[i + 1]));
        total += t;
        if (i / 2 < prof_recs_.size()) prof_recs_[i / 2].ms = t;
    }
    prof_detail.swap(prof_recs_);
    prof_recs_.clear();
    *launches = prof_launches_;
    *ms = total;
    *flops = prof_flops_;
    ev_used_ = 0;
    prof_flops_ = 0;
    prof_launches_ = 0;
}

// ---- the forward pass ---------------------------------------------------------------------------
// ---- one forward at a time per GPU, across the models that live on it -------------------------------------------------
// Models are meant to be one per GPU, but nothing stops a process from holding several on one device (kx_create_replicas with
// repeated ids, tests).  Their streams are non-blocking, so their kernels would run side by side - and the two-CU recurrence
// does not survive that: its 1024-thread, 134 KB-LDS workgroups need a whole CU free at once, the other model's 256-thread conv
// workgroups refill every slot that frees, and a recurrence's second half can starve until the first half's bounded poll gives
// up (measured: a 1 - 2 s stall, KX_ERR_DEVICE, fall-back to the one-CU kernel; profiles/r04_serve_models_per_gpu.txt).  So the
// forwards of the models of one device take turns: a forward's first launch waits (on the GPU, by an event) for the end of
// the previous forward of ANOTHER model on that device, and the host side queues one forward at a time per device.  With one
// model per device this is one uncontended mutex and one event record per forward.
namespace {
struct DeviceGate {
    std::mutex mu;
    hipEvent_t last = nullptr;   // end of the most recent forward queued on this device
    const void* owner = nullptr; // the model that queued it
};
DeviceGate& device_gate(int dev) {
    static DeviceGate g[KX_MAX_DEVICES];
    if (dev < 0 || dev >= KX_MAX_DEVICES) throw Error(1, "device id outside 0.." + std::to_string(KX_MAX_DEVICES - 1));  // (the Model constructor refuses such ids)
    return g[dev];
}
// KX_DEVICE_TURN=0: the models of one device run their forwards side by side (tests; see kx_model_status for what then happens
// to a starved recurrence)
bool device_turn_on() {
    static const bool on = !(getenv("KX_DEVICE_TURN") && atoi(getenv("KX_DEVICE_TURN")) == 0);
    return on;
}
}  // namespace

struct Model::DeviceTurn {
    Model& m;
    DeviceGate& g;
    std::unique_lock<std::mutex> lk;
    const bool on;
    // (CU-partitioned models never compete for a CU: they do not take turns)
    explicit DeviceTurn(Model& mm) : m(mm), g(device_gate(mm.device)), lk(g.mu, std::defer_lock), on(device_turn_on() && mm.n_parts_ == 1) {
        if (!on) return;
        lk.lock();
        if (g.last && g.owner != &m) KX_HIP(hipStreamWaitEvent(m.main_stream_, g.last, 0));
    }
    ~DeviceTurn() {  // (also on a failed forward: whatever it queued is what the next model has to wait for)
        if (!on) return;
        if (!g.last && hipEventCreateWithFlags(&g.last, hipEventDisableTiming) != hipSuccess) g.last = nullptr;
        if (g.last && hipEventRecord(g.last, m.main_stream_) == hipSuccess) g.owner = &m;
        else g.owner = nullptr;
    }
};

void Model::infer_device(const int64_t* d_ids, int64_t t_stride, const int32_t* lens_host, int B,
                         const float* d_styles, const float* speeds_host, int n_speed, uint64_t seed, uint32_t flags,
                         float* d_audio, int64_t audio_ld, int32_t* d_frames, int64_t* need_ld) {
    KX_REQUIRE(B >= 1 && B <= 4096, "infer: batch must be 1..4096 (empty input is an error)");
    KX_REQUIRE(d_ids && lens_host && d_styles && speeds_host, "infer: null argument");
    KX_REQUIRE(n_speed == 1 || n_speed == B, "infer: n_speed must be 1 or B");
    int Tmax = 0;
    for (int b = 0; b < B; ++b) {
        KX_REQUIRE(lens_host[b] >= 1 && lens_host[b] <= 512, "infer: token count must be 1..512");
        KX_REQUIRE((int64_t)lens_host[b] <= t_stride, "infer: lens[b] exceeds the row stride");
        if (lens_host[b] > Tmax) Tmax = lens_host[b];
    }
    for (int i = 0; i < n_speed; ++i) KX_REQUIRE(speeds_host[i] > 0.f, "infer: speed must be > 0");
    KX_HIP(hipSetDevice(device));
    CuScope cu_scope(cu_count_);  // (grid heuristics of the launchers: this model's CUs)
    using clk = std::chrono::steady_clock;
    const clk::time_point t_enter = clk::now();
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    for (double& v : call_ms_) v = 0.0;
    DeviceTurn turn(*this);  // (until this call has queued its last launch)
    B_ = B;
    Tmax_ = Tmax;
    taps_on_ = (flags & 2u) != 0;
    taps_.clear();
    parts_.clear();
    p1_region_ = false;
    lane_ev_used_ = 0;
    prefix_keys_.clear();  // the flat tile lists belong to one call's lengths
    prefix_used_ = 0;
    // Measured (profiles/r03_lanes_dephase.txt): side-by-side chains take 15 % off the batch-1 step (small grids leave CUs
    // idle: 14.1 -> 11.9 ms), 14 % at batch 4, 6 % at batch 16; at batch 64 every launch fills the chip and they change
    // nothing (125.1 vs 125.1 ms) while the per-launch event timings of the profile mode would overlap.  So: lanes for
    // small batches only.
    // (with the per-launch event timing on, one lane: intervals recorded on overlapping streams would be summed side by side)
    n_lanes_ = prof_on_ ? 1 : (lanes_cfg_ ? lanes_cfg_ : (B <= 32 ? N_LANES : 1));
    hT_.assign(lens_host, lens_host + B);
    hF_.assign(B, 0);
    const int Tp = up4(Tmax);
    const int idx_ld = Tmax * 50;
    n_bs_ = 1104;
    const int noise_off = (flags & 1u) ? 1 : 0;
    if (lstm_exchange_bytes(B) > xchg_cap_) {  // exchange buffers of the two-CU LSTM: grown like the arenas
        KX_HIP(hipStreamSynchronize(stream_));
        for (auto*& p : d_xchg_) {
            if (p) KX_HIP(hipFree(p));
            p = nullptr;
            KX_HIP(hipMalloc((void**)&p, lstm_exchange_bytes(B)));
            KX_HIP(hipMemsetAsync(p, 0, lstm_exchange_bytes(B), stream_));  // (stream-ordered before the first recurrence)
        }
        xchg_cap_ = lstm_exchange_bytes(B);
        xchg_epoch_[0] = xchg_epoch_[1] = 0;
    }

    // ===== front half: everything on the token axis ==========================================
    float *emb, *h, *qkv, *ctx, *av, *ff, *dcat, *gxT, *gxT2, *xl, *logits, *te0, *te1, *t_en, *d_speeds;
    int *dur, *idx;
    auto planT = [&](Arena& A) {
        A.off = 0;
        dT_ = A.i(B);
        dF_ = A.i(B);
        d_bad_id_ = reinterpret_cast<unsigned*>(A.i(1));
        d_speeds = A.f(B);
        dur = A.i((size_t)B * 512);
        idx = A.i((size_t)B * idx_ld);
        gb_ = A.f((size_t)B * gb_total_);
        nmean_ = A.f((size_t)B * n_bs_);
        nscale_ = A.f((size_t)B * n_bs_);
        nshift_ = A.f((size_t)B * n_bs_);
        lanes_[0].nmean = nmean_;
        lanes_[0].nscale = nscale_;
        lanes_[0].nshift = nshift_;
        for (int i = 1; i < N_LANES; ++i) {
            lanes_[i].nmean = A.f((size_t)B * n_bs_);
            lanes_[i].nscale = A.f((size_t)B * n_bs_);
            lanes_[i].nshift = A.f((size_t)B * n_bs_);
        }
        const size_t bt = (size_t)B * Tp;
        emb = A.f(bt * 128);
        h = A.f(bt * 768);
        qkv = A.f(bt * 2304);
        ctx = A.f(bt * 768);
        av = A.f(bt * 768);
        ff = A.f(bt * 2048);
        dcat = A.f(bt * 640);
        gxT = A.f(bt * 2048);
        gxT2 = A.f(bt * 2048);  // LSTM input products of the TextEncoder branch (side stream)
        xl = A.f(bt * 512);
        logits = A.f(bt * 50);
        te0 = A.f(bt * 512);
        te1 = A.f(bt * 512);
        t_en = A.f(bt * 512);
    };
    arenaT_.measure = true;
    planT(arenaT_);
    const size_t needT = arenaT_.off;
    arenaT_.measure = false;
    ensure_arena(arenaT_, needT);
    planT(arenaT_);

    KX_HIP(hipMemcpyAsync(dT_, lens_host, B * sizeof(int), hipMemcpyHostToDevice, stream_));
    KX_HIP(hipMemsetAsync(d_bad_id_, 0, sizeof(unsigned), stream_));
    KX_HIP(hipMemcpyAsync(d_speeds, speeds_host, n_speed * sizeof(float), hipMemcpyHostToDevice, stream_));
    const LenMap LT{dT_, 1, 0};
    auto TT = [&](float* p, int C) {
        T t;
        t.p = p; t.bs = (long)C * Tp; t.ld = Tp; t.C = C; t.len = LT; t.Lmax = Tmax;
        return t;
    };
    launch_style_fc(fc_dev_, (int)fc_host_.size(), d_styles, gb_, gb_total_, B, stream_);

    // --- TextEncoder (embedding, 3 x conv k5 + LayerNorm + LeakyReLU, biLSTM) ---
    // Independent of the ALBERT / duration branch below: it runs on the side stream beside it (at small batch
    // neither branch fills the chip: the recurrences use one CU per utterance and direction).
    KX_HIP(hipEventRecord(ev_fork_, stream_));
    KX_HIP(hipStreamWaitEvent(stream2_, ev_fork_, 0));
    struct StreamSwap {  // every launch helper issues on stream_; put the side stream there for this block
        hipStream_t &a, &b;
        StreamSwap(hipStream_t& x, hipStream_t& y) : a(x), b(y) { std::swap(a, b); }
        ~StreamSwap() { std::swap(a, b); }
    };
    T t_ten = TT(t_en, 512);
    {
    StreamSwap on_side(stream_, stream2_);
    T t_te0 = TT(te0, 512), t_te1 = TT(te1, 512);
    launch_embed(d_ids, t_stride, wt("text_encoder.embedding.weight"), 512, te0, t_te0.bs, Tp, dT_, B, Tmax, n_vocab_,
                 d_bad_id_, stream_);
    T* cur = &t_te0;
    T* nxt = &t_te1;
    for (int i = 0; i < 3; ++i) {
        ConvOpts o;
        o.pad = 2;
        conv(convs_.at("text_encoder.cnn." + std::to_string(i)), *cur, *nxt, o);
        const std::string ln = "text_encoder.cnn." + std::to_string(i) + ".1.";
        launch_layernorm_ch(nxt->p, nxt->p, nxt->bs, Tp, 512, LT, B, Tmax, 1e-5f, LN_AFFINE, wt(ln + "gamma"),
                            wt(ln + "beta"), 0, 0.2f, stream_);
        std::swap(cur, nxt);
    }
    tap("text_enc.cnn", *cur);
    lstm(lstms_.at("text_encoder.lstm"), *cur, t_ten, gxT2);
    tap("text_enc.out", t_ten);
    KX_HIP(hipEventRecord(ev_join_, stream_));
    }

    // --- PL-BERT (ALBERT, 12 passes over one shared layer) ---
    const std::string E = "bert.embeddings.";
    const std::string AL = "bert.encoder.albert_layer_groups.0.albert_layers.0.";
    T t_emb = TT(emb, 128), t_h = TT(h, 768), t_qkv = TT(qkv, 2304), t_ctx = TT(ctx, 768), t_a = TT(av, 768),
      t_f = TT(ff, 2048);
    launch_albert_embed(d_ids, t_stride, wt(E + "word_embeddings.weight"), wt(E + "token_type_embeddings.weight"),
                        wt(E + "position_embeddings.weight"), emb, t_emb.bs, Tp, dT_, B, Tmax, n_vocab_, d_bad_id_, stream_);
    launch_layernorm_ch(emb, emb, t_emb.bs, Tp, 128, LT, B, Tmax, 1e-12f, LN_AFFINE, wt(E + "LayerNorm.weight"),
                        wt(E + "LayerNorm.bias"), 0, 0.f, stream_);
    tap("bert.emb", t_emb);
    conv(convs_.at("bert.map"), t_emb, t_h, ConvOpts{});
    for (int l = 0; l < 12; ++l) {
        conv(convs_.at("bert.qkv"), t_h, t_qkv, ConvOpts{});
        launch_attention(qkv, t_qkv.bs, Tp, ctx, t_ctx.bs, Tp, dT_, B, Tmax, stream_);
        ConvOpts od;
        od.resid = &t_h;
        conv(convs_.at("bert.dense"), t_ctx, t_a, od);
        launch_layernorm_ch(av, av, t_a.bs, Tp, 768, LT, B, Tmax, 1e-12f, LN_AFFINE,
                            wt(AL + "attention.LayerNorm.weight"), wt(AL + "attention.LayerNorm.bias"), 0, 0.f, stream_);
        ConvOpts of;
        of.epi = EPI_GELU_NEW;
        conv(convs_.at("bert.ffn"), t_a, t_f, of);
        ConvOpts oo;
        oo.resid = &t_a;
        conv(convs_.at("bert.ffn_out"), t_f, t_h, oo);
        launch_layernorm_ch(h, h, t_h.bs, Tp, 768, LT, B, Tmax, 1e-12f, LN_AFFINE,
                            wt(AL + "full_layer_layer_norm.weight"), wt(AL + "full_layer_layer_norm.bias"), 0, 0.f,
                            stream_);
        if (l == 0) tap("bert.layer0", t_h);
    }
    tap("bert.out", t_h);

    // --- bert_encoder + DurationEncoder (3 x biLSTM + AdaLayerNorm) + duration head ---
    T t_dcat = TT(dcat, 640);
    T t_d512 = t_dcat.rows(0, 512);
    conv(convs_.at("bert_encoder"), t_h, t_d512, ConvOpts{});
    tap("d_en", t_d512);
    launch_fill_style_rows(dcat, t_dcat.bs, Tp, 512, d_styles, 128, dT_, B, Tmax, stream_);
    for (int i = 0; i < 3; ++i) {
        lstm(lstms_.at("predictor.text_encoder.lstms." + std::to_string(2 * i)), t_dcat, t_d512, gxT);
        const float* g = gb_ + fc_off("dur_enc." + std::to_string(i));
        launch_layernorm_ch(dcat, dcat, t_dcat.bs, Tp, 512, LT, B, Tmax, 1e-5f, LN_ADA, g, g + 512, (int)gb_total_, 0.f,
                            stream_);
        tap(("dur_enc." + std::to_string(i)).c_str(), t_dcat);
    }
    T t_xl = TT(xl, 512), t_logits = TT(logits, 50);
    lstm(lstms_.at("predictor.lstm"), t_dcat, t_xl, gxT);
    tap("dur.lstm", t_xl);
    conv(convs_.at("duration_proj"), t_xl, t_logits, ConvOpts{});
    launch_duration(logits, t_logits.bs, Tp, d_speeds, n_speed, dT_, d_pinned_, n_pinned_, dur, dF_, idx, idx_ld, B,
                    stream_);
    KX_HIP(hipMemcpyAsync(hF_.data(), dF_, B * sizeof(int), hipMemcpyDeviceToHost, stream_));
    h_bad_id_ = 0;
    KX_HIP(hipMemcpyAsync(&h_bad_id_, d_bad_id_, sizeof(unsigned), hipMemcpyDeviceToHost, stream_));

    // ===== the one host round trip: predicted frame counts size everything downstream =========
    KX_HIP(hipStreamWaitEvent(stream_, ev_join_, 0));  // the TextEncoder branch joins here
    call_ms_[0] = ms_since(t_enter);  // host time to queue the front half
    KX_HIP(hipStreamSynchronize(stream_));
    call_ms_[1] = ms_since(t_enter);  // ... until the GPU has finished it (the forward's one host wait)
    check_dev_err();
    if (h_bad_id_) {  // a device-side id outside the embedding tables (clamped for the gather, never read out of bounds)
        const unsigned w = h_bad_id_ - 1;
        throw Error(1, "infer: token id outside 0.." + std::to_string(n_vocab_ - 1) + " (utterance " + std::to_string(w >> 16) +
                           ", position " + std::to_string(w & 0xffffu) + ")");
    }
    int Fmax = 0;
    for (int b = 0; b < B; ++b) Fmax = hF_[b] > Fmax ? hF_[b] : Fmax;
    Fmax_ = Fmax;
    if (need_ld) *need_ld = (int64_t)600 * Fmax;
    if (d_frames) KX_HIP(hipMemcpyAsync(d_frames, dF_, B * sizeof(int), hipMemcpyDeviceToDevice, stream_));
    if (audio_ld < (int64_t)600 * Fmax || !d_audio)
        throw Error(1, "infer: audio buffer too small, need ld >= " + std::to_string((long long)600 * Fmax));

    // ===== back half: frame axis ==============================================================
    const int F1p = up4(Fmax), F2p = up4(2 * Fmax), F20p = up4(20 * Fmax), F120p = up4(120 * Fmax + 1);
    const LenMap LF1{dF_, 1, 0}, LF2{dF_, 2, 0}, LF20{dF_, 20, 0}, LF120{dF_, 120, 0}, LF121{dF_, 120, 1};
    auto mk = [&](Arena& A, int C, int ld, LenMap len, int Lmax) {
        T t;
        t.p = A.f((size_t)B * C * ld);
        t.bs = (long)C * ld; t.ld = ld; t.C = C; t.len = len; t.Lmax = Lmax;
        return t;
    };
    auto back = [&](Arena& A) {
        A.off = 0;
        stats_arena_ = &A;
        img_arena_ = &A;
        auto F1 = [&](int C) { return mk(A, C, F1p, LF1, Fmax); };
        auto F2 = [&](int C) { return mk(A, C, F2p, LF2, 2 * Fmax); };
        auto F20 = [&](int C) { return mk(A, C, F20p, LF20, 20 * Fmax); };
        auto F121 = [&](int C) { return mk(A, C, F120p, LF121, 120 * Fmax + 1); };
        // --- alignment expand + shared biLSTM + F0 / N predictors (ProsodyPredictor.F0Ntrain) ---
        T en = F1(640);
        if (!dry_) launch_gather_cols(dcat, t_dcat.bs, Tp, en.p, en.bs, en.ld, 640, idx, idx_ld, dF_, B, Fmax, stream_);
        float* gxF = A.f((size_t)B * Fmax * 2048);
        T xsh = F1(512);
        lstm(lstms_.at("predictor.shared"), en, xsh, gxF);
        tap("pred.shared", xsh);
        T curves = F2(2);  // row 0 = F0 curve, row 1 = N curve, length 2F
        // The F0 and the N branch read xsh and are independent: N goes to lane 1, F0 stays here.  The raw InstanceNorm sums
        // of xsh are computed once, before the fork (stats() caches them per tensor).
        stats(xsh, "predictor.F0.0.norm1");
        hipEvent_t ev_n = nullptr;
        for (int br = 1; br >= 0; --br) {
            const std::string P = std::string("predictor.") + (br == 0 ? "F0" : "N");
            LaneScope on_lane(*this, br);
            T y0 = F1(512);
            adain_resblk(P + ".0", xsh, y0, false, A.f((size_t)B * 512 * F1p), nullptr, nullptr);
            T y1 = F2(256);
            float* wa = A.f((size_t)B * 256 * F2p);
            float* wb = A.f((size_t)B * 256 * F2p);
            float* wc = A.f((size_t)B * 512 * F2p);
            adain_resblk(P + ".1", y0, y1, true, wa, wb, wc);
            T y2 = F2(256);
            adain_resblk(P + ".2", y1, y2, false, A.f((size_t)B * 256 * F2p), nullptr, nullptr);
            conv(convs_.at(P + "_proj"), y2, curves.rows(br, 1), ConvOpts{});
            if (br == 1) ev_n = record_here();
        }
        wait_here(ev_n);
        tap("pred.F0", curves.rows(0, 1));
        tap("pred.N", curves.rows(1, 1));
        // --- Generator, source side: harmonic source -> STFT -> noise_convs / noise_res of both stages.  It depends on the
        // F0 curve only, so it runs on a lane of its own beside the decoder and the first generator stage.
        const std::string G = "decoder.generator.";
        T ns[2];
        hipEvent_t ev_ns[2] = {nullptr, nullptr};
        size_t part_n[2];
        p1_region_ = true;  // (from here on: generator and decoder convs)
        {
            LaneScope on_lane(*this, 3);
            const long hs_ld = (long)600 * Fmax;
            float* har_src = A.f((size_t)B * hs_ld);
            float* phase = A.f((size_t)B * 9 * 2 * Fmax);
            if (!dry_)
                launch_source(curves.p, curves.bs, dF_, B, Fmax, wt(G + "m_source.l_linear.weight"),
                              wt(G + "m_source.l_linear.bias"), seed, utt_base, d_utt_seeds_, noise_off, phase, har_src, hs_ld, stream_);
            if (taps_on_ && !dry_) {
                T hs;
                hs.p = har_src; hs.bs = hs_ld; hs.ld = (int)hs_ld; hs.C = 1; hs.len = LenMap{dF_, 600, 0}; hs.Lmax = 600 * Fmax;
                tap("gen.har_source", hs);
            }
            T har = F121(22);
            if (!dry_) launch_stft(har_src, hs_ld, har.p, har.bs, har.ld, dF_, B, Fmax, stft_variant, stream_);
            tap("gen.har", har);
            for (int st = 0; st < 2; ++st) {
                const int ch = st == 0 ? 256 : 128;
                auto S = [&](int C) { return st == 0 ? F20(C) : F121(C); };
                ns[st] = S(ch);
                T t1 = S(ch);
                part_n[st] = (size_t)B * ch * ((st == 0 ? 20 * Fmax : 120 * Fmax + 1) / 64 + 4);  // >= tiles * WN
                float2* part_t1 = static_cast<float2*>(A.alloc(part_n[st] * sizeof(float2)));
                float2* part_xj = static_cast<float2*>(A.alloc(part_n[st] * sizeof(float2)));
                {
                    ConvOpts o;
                    if (st == 0) { o.stride = 6; o.pad = 3; }
                    o.stat_part = part_xj;
                    conv(convs_.at(G + "noise_convs." + std::to_string(st)), har, ns[st], o);
                }
                adain_resblock1(G + "noise_res." + std::to_string(st), st == 0 ? 7 : 11, ns[st], ns[st], t1, ns[st], 0, 1.f,
                                part_t1, part_xj);
                tap(("gen.x_source." + std::to_string(st)).c_str(), ns[st]);
                ev_ns[st] = record_here();
            }
        }
        // --- Decoder (istftnet.py Decoder.forward) ---
        T xcat0 = F1(514);
        if (!dry_)
            launch_gather_cols(t_en, t_ten.bs, Tp, xcat0.p, xcat0.bs, xcat0.ld, 512, idx, idx_ld, dF_, B, Fmax, stream_);
        {
            ConvOpts o;
            o.stride = 2;
            o.pad = 1;
            conv(convs_.at("decoder.F0_conv"), curves.rows(0, 1), xcat0.rows(512, 1), o);
            conv(convs_.at("decoder.N_conv"), curves.rows(1, 1), xcat0.rows(513, 1), o);
        }
        T catA = F1(1090), catB = F1(1090);
        float* wa = A.f((size_t)B * 1024 * F1p);
        float* wb = A.f((size_t)B * 1024 * F1p);
        adain_resblk("decoder.encode", xcat0, catA.rows(0, 1024), false, wa, wb, nullptr);
        tap("dec.encode", catA.rows(0, 1024));
        conv(convs_.at("decoder.asr_res"), xcat0.rows(0, 512), catA.rows(1024, 64), ConvOpts{});
        if (!dry_) {
            launch_copy_rows(xcat0.rows(512, 2).p, xcat0.bs, xcat0.ld, catA.rows(1088, 2).p, catA.bs, catA.ld, 2, LF1, B,
                             Fmax, stream_);
            launch_copy_rows(catA.rows(1024, 66).p, catA.bs, catA.ld, catB.rows(1024, 66).p, catB.bs, catB.ld, 66, LF1,
                             B, Fmax, stream_);
        }
        T* ci = &catA;
        T* co = &catB;
        for (int i = 0; i < 3; ++i) {
            adain_resblk("decoder.decode." + std::to_string(i), *ci, co->rows(0, 1024), false, wa, wb, nullptr);
            tap(("dec.decode." + std::to_string(i)).c_str(), co->rows(0, 1024));
            std::swap(ci, co);
        }
        T g0 = F2(512);
        {
            float* ua = A.f((size_t)B * 512 * F2p);
            float* ub = A.f((size_t)B * 512 * F2p);
            float* uc = A.f((size_t)B * 1090 * F2p);
            adain_resblk("decoder.decode.3", *ci, g0, true, ua, ub, uc);
        }
        tap("dec.decode.3", g0);
        // --- Generator: 2 up-sampling stages -> iSTFT head (the harmonic source / noise path was issued above) ---
        T x = g0;
        for (int st = 0; st < 2; ++st) {
            const int ch = st == 0 ? 256 : 128;
            auto S = [&](int C) { return st == 0 ? F20(C) : F121(C); };
            T xu = S(ch), xs = S(ch);
            wait_here(ev_ns[st]);
            {
                ConvOpts o;  // x = ups(leaky_relu(x, 0.1)) (+ reflection pad on the last stage) + x_source
                o.act = ACT_LEAKY; o.slope = 0.1f; o.pad = 1;
                o.store = ST_UPSCATTER;
                o.up_pad = st == 0 ? 5 : 3;
                o.up_off = st == 0 ? 0 : 1;
                o.up_reflect = st == 0 ? 0 : 1;
                o.up_len = st == 0 ? LF20 : LF120;
                o.resid = &ns[st];
                conv(convs_.at(G + "ups." + std::to_string(st)), x, xu, o);
            }
            tap(("gen.ups." + std::to_string(st)).c_str(), xu);
            // The three resblocks (k = 3, 7, 11) read xu and are averaged: three independent chains, each on a lane of its
            // own with its own scratch; only the last conv of a chain touches the shared running sum xs, in the fixed
            // order k = 3, 7, 11 (events), so the result does not depend on how the chains interleave.  The raw
            // InstanceNorm sums of xu are computed once, here, before the chains fork (stats() caches them per tensor).
            static const int ks[3] = {3, 7, 11};
            const std::string RB = G + "resblocks.";
            stats(xu, RB + std::to_string(st * 3 + 2) + ".adain1.0");
            hipEvent_t ev_r = nullptr;
            for (int j = 0; j < 3; ++j) {
                T xj = S(ch), t1 = S(ch);
                float2* p_t1 = static_cast<float2*>(A.alloc(part_n[st] * sizeof(float2)));
                float2* p_xj = static_cast<float2*>(A.alloc(part_n[st] * sizeof(float2)));
                LaneScope on_lane(*this, j == 2 ? 0 : j + 1);  // (the longest chain stays on the main stream)
                adain_resblock1(RB + std::to_string(st * 3 + j), ks[j], xu, xj, t1, xs, j > 0 ? 1 : 0, j == 2 ? 3.0f : 1.0f,
                                p_t1, p_xj, ev_r);
                if (j < 2) ev_r = record_here();
            }
            tap(("gen.stage." + std::to_string(st)).c_str(), xs);
            x = xs;
        }
        T cp = F121(22);
        {
            ConvOpts o;
            o.act = ACT_LEAKY; o.slope = 0.01f; o.pad = 3;
            conv(convs_.at(G + "conv_post"), x, cp, o);
        }
        p1_region_ = false;
        tap("gen.conv_post", cp);
        float* spec = A.f((size_t)B * 22 * F120p);
        if (!dry_) launch_istft_head(cp.p, cp.bs, cp.ld, spec, d_audio, audio_ld, dF_, B, Fmax, stft_variant, stream_);
        if (taps_on_ && !dry_) {
            T au;
            au.p = d_audio; au.bs = audio_ld; au.ld = (int)audio_ld; au.C = 1; au.len = LenMap{dF_, 600, 0}; au.Lmax = 600 * Fmax;
            tap("audio", au);
        }
    };
    struct ImgArenaScope {  // (pre-split images exist only while the back half is being issued)
        Arena*& p;
        ~ImgArenaScope() { p = nullptr; }
    } img_scope{img_arena_};
    dry_ = true;
    arenaF_.measure = true;
    try {
        back(arenaF_);
    } catch (...) {
        dry_ = false;
        arenaF_.measure = false;
        throw;
    }
    dry_ = false;
    arenaF_.measure = false;
    const size_t needF = arenaF_.off;
    ensure_arena(arenaF_, needF);
    call_ms_[2] = ms_since(t_enter);  // ... until the back half is planned (dry run of the launch sequence)
    try {
        back(arenaF_);
    } catch (...) {
        sync_lanes();  // (nothing of this call may still be running on a side lane when the arenas are handed out again)
        throw;
    }
    call_ms_[3] = ms_since(t_enter);  // ... until the back half is queued (the call returns; the GPU is still running it)
}

// page-locked scratch for the small per-call host arrays (grown when a larger batch arrives; the stream is idle then: a
// call's copies out of it are followed by that call's synchronisations)
int* Model::stage_ints(size_t n) {
    if (n > h_stage_cap_) {
        KX_HIP(hipStreamSynchronize(stream_));
        if (h_stage_) KX_HIP(hipHostFree(h_stage_));
        h_stage_ = nullptr;
        h_stage_cap_ = 0;
        const size_t want = n < 1024 ? 1024 : 2 * n;
        KX_HIP(hipHostMalloc((void**)&h_stage_, want * sizeof(int), hipHostMallocDefault));
        h_stage_cap_ = want;
    }
    return h_stage_;
}

// kx_infer_device: d_ids / d_styles were written by the caller, typically on the legacy null stream (torch's default);
// the model's stream is non-blocking, so that order is made explicit here (an event on the null stream, no host wait).
void Model::order_after_null_stream() {
    KX_HIP(hipSetDevice(device));
    KX_HIP(hipEventRecord(ev_null_, nullptr));
    KX_HIP(hipStreamWaitEvent(stream_, ev_null_, 0));
}

// ... and on the way out: the model's streams are non-blocking, so work the caller queues on the legacy null stream AFTER
// kx_infer_device returns (torch's default stream reading d_audio, or overwriting d_ids / d_styles for the next request) would
// race with the back half that is still queued.  The null stream waits, on the GPU, for the end of this forward: a caller on
// the null stream is ordered as it was when the model's stream was a blocking one; no host stall.  (Callers on other streams
// must use kx_sync: the header says so.)
void Model::order_null_stream_after() {
    if (!ev_done_) KX_HIP(hipEventCreateWithFlags(&ev_done_, hipEventDisableTiming));
    KX_HIP(hipEventRecord(ev_done_, main_stream_));
    KX_HIP(hipStreamWaitEvent(nullptr, ev_done_, 0));
}

void Model::set_voice_table(const float* table, int n_voices) {
    KX_REQUIRE(table && n_voices >= 1 && n_voices <= 4096, "voice table: 1..4096 voices of [511][256] floats");
    KX_HIP(hipSetDevice(device));
    KX_HIP(hipStreamSynchronize(stream_));
    if (d_voices_) {
        for (auto it = owned_.begin(); it != owned_.end(); ++it)
            if (*it == d_voices_) {
                owned_.erase(it);
                break;
            }
        KX_HIP(hipFree(d_voices_));
        d_voices_ = nullptr;
    }
    const size_t n = (size_t)n_voices * 511 * 256;
    d_voices_ = dev_alloc(n);
    KX_HIP(hipMemcpy(d_voices_, table, n * sizeof(float), hipMemcpyHostToDevice));
    n_voices_.store(n_voices, std::memory_order_release);
}

void Model::infer_host(const int64_t* ids, int64_t t_stride, const int32_t* lens, int B, const float* styles,
                       const float* speeds, int n_speed, uint64_t seed, uint32_t flags, float** out,
                       int64_t* out_lens, const uint64_t* utt_seeds) {
    KX_REQUIRE(out && out_lens, "infer: null output argument");
    HostCall hc;
    hc.styles = styles;
    hc.utt_seeds = utt_seeds;
    KX_REQUIRE(styles, "infer: null argument");
    void* p = nullptr;
    *out = nullptr;
    std::vector<int64_t> bytes(B > 0 ? B : 1);
    infer_host_ex(ids, t_stride, lens, B, speeds, n_speed, seed, flags, hc, &p, bytes.data(), out_lens);
    *out = static_cast<float*>(p);
}

// A hand-off time-out of the resident-weights recurrence invalidates the call it happened in, not the request: the host entry
// points run the call once more, now on the streaming recurrence (same bits), and the caller sees a result, not an error.
void Model::infer_host_ex(const int64_t* ids, int64_t t_stride, const int32_t* lens, int B, const float* speeds,
                          int n_speed, uint64_t seed, uint32_t flags, const HostCall& hc, void** out,
                          int64_t* out_bytes, int64_t* out_samples) {
    try {
        infer_host_once(ids, t_stride, lens, B, speeds, n_speed, seed, flags, hc, out, out_bytes, out_samples);
    } catch (const LstmTimeout&) {
        n_rerun_ += 1;
        infer_host_once(ids, t_stride, lens, B, speeds, n_speed, seed, flags, hc, out, out_bytes, out_samples);
    }
    note_clean_forward();
}

void Model::infer_host_once(const int64_t* ids, int64_t t_stride, const int32_t* lens, int B, const float* speeds,
                            int n_speed, uint64_t seed, uint32_t flags, const HostCall& hc, void** out,
                            int64_t* out_bytes, int64_t* out_samples) {
    KX_REQUIRE(out && out_bytes && out_samples, "infer: null output argument");
    *out = nullptr;
    KX_REQUIRE(B >= 1, "infer: empty batch");
    KX_REQUIRE(ids && lens && speeds, "infer: null argument");
    KX_REQUIRE(hc.format >= 0 && hc.format <= 2, "infer: unknown output format");
    const bool by_voice = hc.voice_ids != nullptr;
    KX_REQUIRE(by_voice || hc.styles, "infer: styles or voice ids are required");
    KX_REQUIRE(!hc.kinds || (by_voice && hc.styles), "infer: per-utterance kinds need both styles and voice ids");
    auto kind_of = [&](int b) { return hc.kinds ? hc.kinds[b] : (by_voice ? (hc.max_mix == 1 ? 1 : 2) : 0); };
    auto format_of = [&](int b) { return hc.formats ? hc.formats[b] : hc.format; };
    auto bps_of = [&](int b) { return format_of(b) == 1 ? 8 : (format_of(b) == 2 ? 2 : 4); };
    if (by_voice) {
        KX_REQUIRE(d_voices_ && hc.weights && hc.max_mix >= 1 && hc.max_mix <= 16, "infer: voice table not set or bad mix");
    }
    for (int b = 0; b < B; ++b) {
        KX_REQUIRE(lens[b] >= 1 && lens[b] <= 512 && lens[b] <= t_stride, "infer: token count must be 1..512");
        for (int t = 0; t < lens[b]; ++t) {
            const int64_t id = ids[b * t_stride + t];
            KX_REQUIRE(id >= 0 && id < n_vocab_, "infer: token id outside 0..177");
        }
        KX_REQUIRE(kind_of(b) >= 0 && kind_of(b) <= 2 && format_of(b) >= 0 && format_of(b) <= 2, "infer: unknown kind / output format");
        if (kind_of(b) != 0) {
            KX_REQUIRE(lens[b] >= 2, "infer: voice rows need the two 0 pads (row = tokens - 2)");
            bool any = false;
            for (int k = 0; k < hc.max_mix; ++k) {
                const int v = hc.voice_ids[(size_t)b * hc.max_mix + k];
                KX_REQUIRE(v < n_voices_, "infer: voice id outside the table");
                any = any || v >= 0;
            }
            KX_REQUIRE(any && (kind_of(b) != 1 || hc.voice_ids[(size_t)b * hc.max_mix] >= 0), "infer: no voice given");
        }
    }
    KX_HIP(hipSetDevice(device));
    // I/O staging lives in its own arena: ids, styles, frames, noise keys, voice picks, audio, packed audio
    int64_t* d_ids;
    float* d_styles;
    int* d_fr;
    uint64_t* d_seeds;
    int *d_vid, *d_rows, *d_kinds, *d_formats;
    float* d_w;
    void* d_packed;
    long* d_off;
    struct SeedGuard {  // the per-utterance key pointer is valid only during this call
        const uint64_t*& p;
        ~SeedGuard() { p = nullptr; }
    } seed_guard{d_utt_seeds_};
    const int mm = by_voice ? hc.max_mix : 1;
    int bytes_per_sample = 0;  // (the widest form of the batch sizes the packed buffer)
    for (int b = 0; b < B; ++b) bytes_per_sample = bps_of(b) > bytes_per_sample ? bps_of(b) : bytes_per_sample;
    auto planIO = [&](Arena& A, size_t audio_floats) {
        A.off = 0;
        d_ids = static_cast<int64_t*>(A.alloc((size_t)B * t_stride * 8));
        d_seeds = static_cast<uint64_t*>(A.alloc((size_t)B * 8));
        d_styles = A.f((size_t)B * 256);
        d_fr = A.i(B);
        d_vid = A.i((size_t)B * mm);
        d_rows = A.i(B);
        d_kinds = A.i(B);
        d_formats = A.i(B);
        d_w = A.f((size_t)B * mm);
        d_off = static_cast<long*>(A.alloc((size_t)B * 8));
        d_packed = A.alloc(audio_floats * bytes_per_sample);  // compact output: utterances back to back
        return A.f(audio_floats);
    };
    // worst case length is 50 frames per token; start from a typical 8 and retry once if short
    int Tmax = 0;
    for (int b = 0; b < B; ++b) Tmax = lens[b] > Tmax ? lens[b] : Tmax;
    int64_t ld = (int64_t)600 * Tmax * 8;
    for (int attempt = 0; attempt < 2; ++attempt) {
        arenaIO_.measure = true;
        planIO(arenaIO_, (size_t)B * ld);
        const size_t need = arenaIO_.off;
        arenaIO_.measure = false;
        ensure_arena(arenaIO_, need);
        float* d_audio = planIO(arenaIO_, (size_t)B * ld);
        KX_HIP(hipMemcpyAsync(d_ids, ids, (size_t)B * t_stride * 8, hipMemcpyHostToDevice, stream_));
        if (hc.styles)  // (explicit rows first: the mix kernel then fills the rows of the utterances that name voices)
            KX_HIP(hipMemcpyAsync(d_styles, hc.styles, (size_t)B * 256 * 4, hipMemcpyHostToDevice, stream_));
        if (by_voice) {
            std::vector<int> rows(B);
            for (int b = 0; b < B; ++b) rows[b] = lens[b] >= 2 ? lens[b] - 2 : 0;  // tokens before the 0 padding (koko.rs:1161-1166)
            KX_HIP(hipMemcpyAsync(d_vid, hc.voice_ids, (size_t)B * mm * 4, hipMemcpyHostToDevice, stream_));
            KX_HIP(hipMemcpyAsync(d_w, hc.weights, (size_t)B * mm * 4, hipMemcpyHostToDevice, stream_));
            int* st = stage_ints((size_t)3 * B);  // page-locked: rows | kinds | formats, copied on the model's own stream
            memcpy(st, rows.data(), (size_t)B * 4);
            KX_HIP(hipMemcpyAsync(d_rows, st, (size_t)B * 4, hipMemcpyHostToDevice, stream_));
            if (hc.kinds) {
                memcpy(st + B, hc.kinds, (size_t)B * 4);
                KX_HIP(hipMemcpyAsync(d_kinds, st + B, (size_t)B * 4, hipMemcpyHostToDevice, stream_));
            }
            launch_style_mix(d_voices_, n_voices_, d_vid, d_w, mm, d_rows, hc.kinds ? d_kinds : nullptr, d_styles, B, stream_);
        }
        d_utt_seeds_ = nullptr;
        if (hc.utt_seeds) {
            KX_HIP(hipMemcpyAsync(d_seeds, hc.utt_seeds, (size_t)B * 8, hipMemcpyHostToDevice, stream_));
            d_utt_seeds_ = d_seeds;
        }
        int64_t need_ld = 0;
        try {
            infer_device(d_ids, t_stride, lens, B, d_styles, speeds, n_speed, seed, flags, d_audio, ld, d_fr, &need_ld);
        } catch (const Error& e) {
            if (attempt == 0 && need_ld > ld) {
                ld = need_ld;
                continue;
            }
            throw;
        }
        // frame counts are known (the forward's one host sync): pack the B waveforms back to back on the GPU in the
        // requested sample format, then ONE asynchronous copy into a page-locked host buffer
        std::vector<long>& off = h_off_;
        off.assign(B, 0);
        int64_t total = 0;
        for (int b = 0; b < B; ++b) {
            out_samples[b] = (int64_t)600 * hF_[b];
            out_bytes[b] = out_samples[b] * bps_of(b);
            off[b] = (long)total;
            total += out_bytes[b];
        }
        KX_HIP(hipMemcpyAsync(d_off, off.data(), (size_t)B * 8, hipMemcpyHostToDevice, stream_));
        if (hc.formats) {
            int* st = stage_ints((size_t)3 * B) + 2 * (size_t)B;
            memcpy(st, hc.formats, (size_t)B * 4);
            KX_HIP(hipMemcpyAsync(d_formats, st, (size_t)B * 4, hipMemcpyHostToDevice, stream_));
        }
        launch_pack_audio(d_audio, ld, dF_, B, Fmax_, hc.format, d_packed, 0, d_off, stream_, hc.formats ? d_formats : nullptr);
        char* host = static_cast<char*>(host_out_alloc((size_t)(total > 0 ? total : 1)));
        hipError_t e = hipMemcpyAsync(host, d_packed, (size_t)total, hipMemcpyDeviceToHost, stream_);
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);
        if (e != hipSuccess) {
            host_out_free(host);
            throw Error(3, std::string("infer: D2H copy failed: ") + hipGetErrorString(e));
        }
        // the sticky device error word once more: kernels of the back half (the frame-axis LSTM) can raise it after the
        // forward's mid-way check, and the audio of THIS call would be garbage -- it must fail here, not in the next call
        try {
            check_dev_err();
        } catch (...) {
            host_out_free(host);
            throw;
        }
        *out = host;
        return;
    }
}


// ---- pooled page-locked host buffers for the results ------------------------------------------------------
namespace {
struct HostPool {
    std::mutex mu;
    std::map<void*, size_t> cap;             // every live pinned buffer -> capacity
    std::multimap<size_t, void*> free_list;  // idle ones by capacity
    std::map<void*, void*> alias;            // pointer handed to an owner -> the shared buffer it lies in (host_out_share)
    std::map<void*, int> refs;               // shared buffer -> owners still holding a part
    size_t idle_bytes = 0;
    size_t live_shared = 0;                  // capacity of the shared buffers in `refs`
    static constexpr size_t kMaxIdle = size_t(1) << 30;
    ~HostPool() {
        for (auto& kv : free_list) (void)hipHostFree(kv.second);
    }
};
HostPool& host_pool() {
    static HostPool* p = new HostPool;  // (leaked on purpose: buffers may outlive static destruction order)
    return *p;
}
}  // namespace

void* host_out_alloc(size_t bytes) {
    HostPool& P = host_pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.free_list.lower_bound(bytes);
        if (it != P.free_list.end() && it->first <= 2 * bytes + (1 << 20)) {
            void* p = it->second;
            P.idle_bytes -= it->first;
            P.free_list.erase(it);
            return p;
        }
    }
    const size_t want = (bytes + (1 << 20) - 1) & ~((size_t(1) << 20) - 1);
    void* p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess || !p) {
        (void)hipGetLastError();
        p = malloc(bytes);  // pageable memory still works with hipMemcpyAsync (staged by the runtime)
        if (!p) throw Error(3, "infer: out of host memory");
        return p;
    }
    std::lock_guard<std::mutex> lk(P.mu);
    P.cap[p] = want;
    return p;
}

void host_out_share(void* base, void* const* parts, int n) {
    if (!base || n <= 0) return;
    HostPool& P = host_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    // one reference per DISTINCT pointer: two parts with the same address (a zero-byte part; cannot happen today, an utterance
    // has at least one frame) would share one key, and a count of n would then never come down to zero
    int distinct = 0;
    for (int i = 0; i < n; ++i) distinct += P.alias.emplace(parts[i], base).second ? 1 : 0;
    P.refs[base] = distinct;
    auto it = P.cap.find(base);
    if (it != P.cap.end()) P.live_shared += it->second;
}

size_t host_out_live_bytes() {
    HostPool& P = host_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    return P.live_shared;
}

void host_out_free(void* p) {
    if (!p) return;
    HostPool& P = host_pool();
    size_t c = 0;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto al = P.alias.find(p);
        if (al != P.alias.end()) {  // one part of a shared batch buffer: the buffer itself goes when the last part has gone
            void* base = al->second;
            P.alias.erase(al);
            auto rf = P.refs.find(base);
            if (rf != P.refs.end() && --rf->second > 0) return;
            if (rf != P.refs.end()) P.refs.erase(rf);
            p = base;
            auto cb = P.cap.find(base);
            if (cb != P.cap.end()) P.live_shared -= cb->second < P.live_shared ? cb->second : P.live_shared;
        }
        auto it = P.cap.find(p);
        if (it != P.cap.end()) {
            c = it->second;
            if (P.idle_bytes + c <= HostPool::kMaxIdle) {
                P.free_list.emplace(c, p);
                P.idle_bytes += c;
                return;
            }
            P.cap.erase(it);
        }
    }
    if (c) (void)hipHostFree(p);
    else free(p);
}

}  // namespace kx
