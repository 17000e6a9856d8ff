/*
 * kokorox_hip.h — C ABI of libkokorox_hip.so: the MI355X-native replacement for the one
 * hot path of byteowlz/kokorox, the Kokoro-82M forward pass that the reference runs as
 * `ort::Session::run()` inside `OrtKoko::infer`.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference checkout).  Plain pointers and sizes only; no C++/torch types; nothing
 * throws or unwinds across this boundary: every call returns a status and the text of
 * the last failure is available from kx_last_error().
 *
 * Threading: a kx_model may be shared by any number of OS threads.  Calls on one model
 * are serialised on an internal mutex, which is exactly what the reference's
 * `Mutex<Session>` does (kokorox/src/onn/ort_koko.rs:14,78).  Use one model per GPU.
 */
#ifndef KOKOROX_HIP_H
#define KOKOROX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KX_OK 0
#define KX_ERR_INVALID 1   /* bad argument (empty batch, id out of range, T > 512 ...)   */
#define KX_ERR_IO 2        /* weight file missing / malformed                             */
#define KX_ERR_DEVICE 3    /* HIP runtime failure, no gfx950 device                       */
#define KX_ERR_STATE 4     /* unknown tap name, profiling not enabled ...                 */

#define KX_STYLE_DIM 256       /* ort_koko.rs:44 "1,256"; koko.rs:1255-1306 builds the row */
#define KX_SAMPLES_PER_FRAME 600
#define KX_SAMPLE_RATE 24000
#define KX_MAX_TOKENS 512      /* model position table; koko.rs:783 chunks to <=500 + 2 pads */

/* kx_infer flags */
#define KX_FLAG_NOISE_OFF 1u   /* zero the SineGen additive noise (timing experiments only) */
#define KX_FLAG_TAPS 2u        /* keep named intermediates for kx_debug_tap (tests)          */

typedef struct kx_model kx_model;

/* Replaces `init_ort(dylib_path)` (kokorox/src/onn/mod.rs:19-49): one-off runtime
 * initialisation.  Checks that `device_id` exists and is a gfx950 part.
 * err/err_len: optional buffer for a message when the return value is not KX_OK. */
int kx_init(int device_id, char* err, size_t err_len);

/* Replaces `OrtKoko::new(model_path)` → `OrtBase::load_model` → `commit_from_file`
 * (kokorox/src/onn/ort_koko.rs:31-35, ort_base.rs:14-39; call site kokorox/src/tts/koko.rs:570-573).
 * `weights_path` is what the reference passes there: the Hugging Face `onnx/model.onnx` (path built at
 * kokorox/src/utils/hf_cache.rs:128-158; its fp16 / int8 / 4-bit siblings of hf_cache.rs:135-144 load as well, their
 * weights de-quantised to f32).  The library reads the ONNX initialisers itself (csrc/onnx_import.cpp: no protobuf
 * package, no ONNX Runtime), places them by the rules of kokorox_amd/importer.py, uploads them and repacks the
 * contraction weights for the MFMA kernels.  A file that starts with the magic "KXHIPW01" is taken as the library's own
 * pre-converted container (kokorox_amd/weights.py) instead; `<path>.kxw` beside an .onnx is used when it is at least as
 * new as the .onnx, and written when the environment has KOKOROX_KXW_CACHE=1.
 * Returns NULL on failure with the reason in err (KX_ERR_IO class: missing / truncated / external-data files, tensors
 * of the model that could not be found — the message lists initialisers that were not placed). */
kx_model* kx_create(const char* weights_path, int device_id, char* err, size_t err_len);

/* Host only (no GPU is touched): `model.onnx` → the KXHIPW01 container kx_create would build from it in memory, written
 * to out_path — for a deployment that wants to convert once (the bytes equal `python -m kokorox_amd.importer`'s).
 * No reference counterpart: ONNX Runtime parses the file on every start (ort_base.rs:27-33). */
int kx_import_onnx(const char* onnx_path, const char* out_path, char* err, size_t err_len);

/* Same, from a weight blob that is ALREADY resident in this GPU's memory, e.g. after the
 * one-time RCCL broadcast over xGMI.  The blob is borrowed for the
 * duration of the call only (the library keeps its own repacked copy and copies the
 * small tensors it needs). */
kx_model* kx_create_from_device_blob(const void* d_blob, size_t n_bytes, int device_id,
                                     char* err, size_t err_len);

/* One model per GPU for the server (the reference's server is ONE process, kokorox-openai/src/lib.rs:
 * 370-439): reads the weight file ONCE (`.onnx` or container, as kx_create), uploads it to device_ids[0] over PCIe and fans it out to the other
 * devices with concurrent device-to-device copies (hipMemcpyPeerAsync, one xGMI link per destination), then builds
 * every model from its resident blob, the models side by side (one host thread each).  out_models[n] receives the handles,
 * ready for kx_dispatcher_create.  A device id that is given k times (k <= 8) gets k CU-PARTITIONED models, each confined to
 * 1 / k of the device's CUs (kx_create_partition): they run their forwards side by side on the one GPU and never compete for a
 * CU (KX_REPLICA_PARTITION=0: whole-device models that take turns instead).  On failure nothing is leaked and every entry of
 * out_models is NULL.
 * (A multi-PROCESS launch — one rank per GPU, as bench.py under torch.distributed — uses an RCCL broadcast of the
 * blob and kx_create_from_device_blob instead: kokorox_amd/dist.py.) */
int kx_create_replicas(const char* weights_path, const int* device_ids, int n, kx_model** out_models, char* err,
                       size_t err_len);
/* Milestones of the process's last kx_create_replicas call, ms from its entry: out3[0] weight file read (and, for an .onnx,
 * converted), [1] blob resident on every device (PCIe upload + the fan-out copies), [2] every model built. */
int kx_replicas_times(double* out3);

/* A model confined to CUs [part, part + 1) x CUs / n_parts of the device (n_parts 1 .. 8; n_parts = 1 is kx_create): every
 * stream of the model carries that CU mask (hipExtStreamCreateWithCUMask; the driver deals the bits over the XCCs, so a
 * partition keeps its share of every XCC's L2).  Models with disjoint partitions run concurrently on one GPU with no
 * contention for CUs -- two forwards in flight per GPU behind one dispatcher -- and give the bits of a whole-device model.
 * No reference counterpart (one Mutex<Session> per process, ort_koko.rs:78). */
kx_model* kx_create_partition(const char* weights_path, int device_id, int part, int n_parts, char* err, size_t err_len);

/* Replaces `Drop for OrtKoko` / TTSKoko::cleanup (kokorox/src/tts/koko.rs:1338-1375). */
void kx_destroy(kx_model* m);

/* Text of the CALLING THREAD's last failed call on this model ("" if its last call succeeded or it never failed):
 * several threads may share one model (calls are serialised inside), and neither another thread's success nor its
 * failure touches what this thread reads.  The pointer stays valid until the calling thread's next call on any model.
 * kx_last_error_copy writes the same text into a caller buffer (always NUL-terminated). */
const char* kx_last_error(const kx_model* m);
int kx_last_error_copy(const kx_model* m, char* buf, size_t buf_len);

/* Replaces `OrtKoko::infer(tokens, styles, speed)` (kokorox/src/onn/ort_koko.rs:37-91;
 * caller kokorox/src/tts/koko.rs:1177).
 *   ids      [B, t_stride] int64, row b holds lens[b] token ids already wrapped with the
 *            0 pad at both ends (koko.rs:1169-1173); ids must be in 0..177 (vocab.rs:5-20)
 *   lens     [B] tokens per utterance incl. the two pads, 1..512 (3..512 for real text).  The reference is
 *            batch-1 (koko.rs:1175); B>1 gives the same result as B separate calls.
 *   styles   [B,256] float32 voice-style rows (mix_styles, koko.rs:1255-1306)
 *   speeds   [n_speed] float32, n_speed = 1 (shared, ort_koko.rs:67-68) or B
 *   seed     Philox key of the harmonic-source noise; utterance b of the call draws the
 *            stream (seed, utt_base + b)
 *   out      *out = a library-owned (pooled, page-locked) float32 buffer holding the B waveforms back to back;
 *            out_lens[b] = samples of utterance b (600 * predicted frames).  Free with
 *            kx_free_audio.  The reference likewise returns an owned copy
 *            (ort_koko.rs:85 `data.to_vec()`).
 * Empty input (B = 0 or a len < 1) is an error, not a panic as in ort_koko.rs:56,61. */
int kx_infer(kx_model* m, const int64_t* ids, int64_t t_stride, const int32_t* lens, int B,
             const float* styles, const float* speeds, int n_speed, uint64_t seed,
             uint32_t flags, float** out, int64_t* out_lens);

void kx_free_audio(float* p);

/* Device-resident form of the same call for batch serving and for bench.py: every
 * pointer is GPU memory of this model's device, nothing crosses PCIe, nothing is
 * allocated per call after the first call of a given shape.
 *   d_audio      [B, audio_ld] float32, written for samples < 600*frames[b]
 *   d_frames     [B] int32 predicted frame counts
 * Returns KX_ERR_INVALID if audio_ld is smaller than the longest waveform (the needed
 * length is then in *need_ld), or if a token id in d_ids lies outside 0..177: device-side ids are range-checked by
 * the embedding kernels (clamped for the gather, so nothing is read out of bounds) and reported at the call's one
 * host synchronisation point.  The call returns after the work has been queued on the
 * model's stream and its frame counts are known; kx_sync waits for completion.
 * Streams: the model's streams are NON-BLOCKING (they do not synchronise with the legacy null stream).  The call orders
 * itself against the NULL stream on both sides, on the GPU, without a host stall: its reads of d_ids / d_styles wait for what
 * the caller had queued on the null stream before the call, and null-stream work queued AFTER the call returns (torch's default
 * stream reading d_audio / d_frames, or writing the next request into d_ids / d_styles) waits for the end of this forward.
 * A caller that produces the inputs or consumes the outputs on any OTHER stream (a torch side stream, a per-thread default
 * stream) is not ordered at all: it must finish its writes before the call and call kx_sync before touching d_audio,
 * d_frames, d_ids or d_styles again. */
int kx_infer_device(kx_model* m, const int64_t* d_ids, int64_t t_stride, const int32_t* lens_host,
                    int B, const float* d_styles, const float* speeds_host, int n_speed,
                    uint64_t seed, uint32_t flags, float* d_audio, int64_t audio_ld,
                    int32_t* d_frames, int64_t* need_ld);

int kx_sync(kx_model* m);

/* What was loaded and how it runs.  out8[0]: the weight file's kind -- 0 the library's KXHIPW01 container, 1 fp32 ONNX
 * (onnx/model.onnx, hf_cache.rs:10), 2 fp16 / bf16 ONNX (model_fp16: widened to f32), 3 an 8-bit quantised ONNX variant
 * (model_quantized / model_uint8 / model_q8f16 / model_uint8f16), 4 a 4-bit one (model_q4 / model_q4f16), -1 a cached
 * conversion, -2 built from a device blob (kind unknown).  out8[1]: 1 = the model computes what the reference computes for
 * this file; 0 for kinds 3 and 4: ONNX Runtime runs those as DynamicQuantizeLinear -> MatMulInteger / ConvInteger /
 * MatMulNBits, i.e. it quantises the ACTIVATIONS at run time, while this library de-quantises the WEIGHTS at load and then
 * runs f32-class arithmetic -- closer to the fp32 model than the reference's own output for that file, and therefore not
 * within 1e-4 of it (a notice is printed at load; hf_cache.rs:135-144, ort_base.rs:27-33).  out8[2]: conv mode
 * (kx_get_conv_mode); [3] rows of the embedding tables; [4] voices in the device table; [5] / [6] CU partition and number of
 * partitions (kx_create_partition; 0 / 1 = the whole device); [7] CUs the model launches on. */
int kx_model_info(kx_model* m, int64_t* out8);

/* What the model's LSTM recurrences are running on, and whether anything went wrong (SURVEY 8b "Threading"; the reference has
 * nothing to report: one Mutex<Session>, ort_koko.rs:78).  out4[0]: 0 = the resident-weights forms (two / four workgroups per
 * utterance and direction, which hand h over between CUs), 1 = the streaming fall-back (one workgroup, no hand-off); out4[1]:
 * hand-off time-outs so far (a part of a recurrence starved behind other work on the GPU for ~0.1 s); out4[2]: clean forwards
 * left until the resident forms return (64 after a time-out, 0 in normal operation); out4[3]: calls that were re-run
 * transparently after a time-out.  Both forms give the SAME BITS, so the switch never changes a result; it costs time
 * (~6 us against 1.6 - 2.5 us per recurrence step).  All zeros in a healthy deployment. */
int kx_model_status(kx_model* m, int64_t* out4);

/* Benchmark control (SURVEY.md §8d "duration pinning"): when set, the predicted duration
 * of token t is replaced by pattern[t % n] for every utterance (the duration head still
 * runs).  n = 0 clears it. */
int kx_set_pinned_durations(kx_model* m, const int32_t* pattern, int n);

/* For servers: one discarded forward of B utterances x n_tokens (incl. the two pads) with every duration pinned to
 * frames_per_token, so that the model's arenas, its page-locked result buffer and its tile tables already have the largest
 * (batch x length) shape the deployment expects when the first request arrives.  Without it the arenas grow on demand - a
 * stream sync + hipFree + hipMalloc of gigabytes, measured as a 1.3 s latency outlier the first time a larger batch shape
 * arrives.  No reference counterpart (ONNX Runtime allocates per run).  A pinned pattern set before stays in force. */
int kx_warmup(kx_model* m, int B, int n_tokens, int frames_per_token);
/* Host-side milestones of the last kx_infer_device / kx_infer* call on this model, in ms from the call's entry:
 * out4[0] front half queued, [1] front half finished on the GPU (the forward's one host wait: the predicted frame counts size
 * everything downstream), [2] back half planned, [3] back half queued (kx_infer_device returns here, the GPU still running). */
int kx_call_times(kx_model* m, double* out4);

/* Capacities in bytes of the three device arenas (token axis, frame axis, I/O): they change only when a larger shape arrives. */
int kx_arena_bytes(kx_model* m, int64_t* out3);

/* Contraction arithmetic of the conv/linear kernel: 0 = f32 MFMA (v_mfma_f32_32x32x2_f32, exact f32
 * products), 1 = f16x3 split MFMA (three v_mfma_f32_32x32x16_f16 per K-step on hi/lo halves, ~22
 * significant bits per product, f32 accumulation; env KOKOROX_CONV=f16x3).  Env KOKOROX_CONV=f32 selects 0 at create.
 * 6 = f16f8, the DEFAULT since round 5 (env KOKOROX_CONV=f16f8): mode 1, except that the snake convs of the generator (3, 7 and 11
 * taps: 85 % of the forward's multiply-adds) carry the two cross terms of the split product, a_lo b_hi + a_hi b_lo, on the 8-bit
 * scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4 on e4m3 images of the four operands) and a_hi b_hi on v_mfma_f32_16x16x32_f16:
 * two MFMA-equivalents per product instead of three, ~17 significant bits per product instead of 22 (measured 1e-5 relative per
 * conv output; the waveform moves by < 1e-5 against mode 1), inside the 1e-4 parity band by the same protocol as modes 0 and 1
 * (tests/test_gpu_forward.py::test_ragged_batch_matches_oracle) and far above the bf16 that BASELINE configs[2] names.  Their
 * snake activation takes sin^2 from the hardware cosine there (3e-6 absolute).  The 8-bit weight images are built at create
 * (or at the first selection of the mode).
 * 4 = reduced precision, opt-in (env KOKOROX_CONV=f16): the decoder / generator convs of the direct-A kernel issue ONE
 * f16 MFMA per product (weights and activations rounded to f16, f32 accumulation) -- the library's counterpart of the
 * reference's `model_fp16` / quantised variants run at their own precision (kokorox/src/utils/hf_cache.rs:135-144) and
 * of BASELINE configs[2] "bf16"; duration head, F0 / N predictor, harmonic source and STFT pair stay f32-class.  Never
 * the default: the waveform leaves the 1e-4 parity band (measured bound in tests/test_gpu_forward.py).
 * 5 = the same on bf16 (env KOKOROX_CONV=bf16; v_mfma_f32_32x32x16_bf16, a bf16 weight image built at the first selection):
 * the dtype BASELINE configs[2] names; 8 significant bits instead of 11, so its error is ~8x that of mode 4 (same test). */
int kx_set_conv_mode(kx_model* m, int mode);
int kx_get_conv_mode(kx_model* m);

/* Which STFT / inverse-STFT pair the generator uses (n_fft 20, hop 5 on both):
 *   0 = the conv-based pair of the ONNX export (upstream custom_stft.py: magnitude sqrt(re^2+im^2+1e-14), phase
 *       atan2 with (im == 0 && re < 0) -> +pi, inverse = cos/sin * window / n_fft transposed convolutions summed as
 *       real - imag, without one-sided doubling or window-envelope division).  Default: `model.onnx` is what the
 *       reference runs (kokorox/src/utils/hf_cache.rs:8-10, ort_koko.rs:79).
 *   1 = torch.stft / torch.istft semantics (the PyTorch checkpoint's own path).
 * Both are restated from the public upstream sources from memory: which of the two the
 * shipped graph contains is the first thing to confirm when a real model.onnx is reachable.  Env KOKOROX_STFT=torch
 * selects 1 at create. */
int kx_set_stft_variant(kx_model* m, int variant);
int kx_get_stft_variant(kx_model* m);

/* Streams ("lanes") of the back half: 1 = every launch on the model's one stream; 4 = the harmonic-source / noise path
 * and the three resblock chains of each generator stage are issued on streams of their own and meet through events;
 * 0 (default) = 4 for batches of up to 32 utterances, where launches leave CUs idle (batch 1: 14.1 -> 11.9 ms), else 1.
 * Also side by side in that mode: the F0 and N predictor branches, and the 1x1 shortcut conv of every AdainResBlk1d.
 * Results are bit-identical for every value: only the last conv of a chain touches the shared running sum, in a fixed
 * order.  Env KX_LANES at create. */
int kx_set_lanes(kx_model* m, int n_lanes);

/* First utterance index used for the noise stream of the next calls (default 0). */
int kx_set_utterance_base(kx_model* m, uint64_t utt_base);

/* Per-kernel-class HIP-event timing on the model's stream (bench.py roofline leg).
 * While enabled every launch of the dominant kernel family -- the 128-row conv / GEMM kernels: conv1d_f16x3_da_kernel,
 * conv1d_f16x3_dag_kernel and conv1d_f16x3_kernel<128,..> (conv1d_mfma_kernel<128,128,2,2> in f32 mode) -- is
 * bracketed by events on the stream it is issued on.
 * kx_profile_read drains them: launches, summed milliseconds and summed algorithmic
 * FLOPs (2*Cout*Cin*k*columns per launch) since the last read. */
int kx_profile_enable(kx_model* m, int on);
int kx_profile_read(kx_model* m, int64_t* launches, double* total_ms, double* total_flops);
/* Per-launch records of the last kx_profile_read: out[i*10 + 0..9] = GEMM rows, Cin, taps, dilation,
 * stride, store form, summed columns, FLOPs, milliseconds, algorithmic HBM bytes (input once + residual /
 * running sum where read + output once + weights once).  out = NULL returns the count only. */
int kx_profile_detail(kx_model* m, double* out, int64_t cap_rows, int64_t* n_rows);
/* Unfused InstanceNorm statistics passes (in_stats_kernel) since the last call: launches and the bytes they must read
 * (each reads its tensor exactly once) - the known byte count the PMC tooling checks FETCH_SIZE against. */
int kx_profile_aux(kx_model* m, int64_t* stats_launches, double* stats_bytes);

/* ---- real-weights readiness (tools/real_weights_report.py) -------------------------------------------------
 * The f16x3 contraction carries every f32 operand as two f16 halves.  That is f32-class for O(1) activations; an input
 * whose magnitude is far below 1e-3 loses its low half to the f16 subnormal quantum, one above 65504 is clamped.
 * kx_diag_enable(1) makes every conv launch of the following kx_infer* calls also measure its input after the AdaIN
 * affine (absmax, rms); kx_diag_count / kx_diag_get read the records (vals7 = GEMM rows, Cin, taps, current
 * pre-scale exponent, absmax, rms, elements).  kx_set_act_prescale(name, e) multiplies that layer's transformed input
 * by 2^e before the split and divides the accumulators by it in the epilogue - exact, so results are unchanged
 * wherever the split was already lossless.  Layer names are the weight names without ".weight" (LSTM input
 * projections: "<lstm>.ih"). */
int kx_diag_enable(kx_model* m, int on);
int kx_diag_count(kx_model* m, int64_t* n);
int kx_diag_get(kx_model* m, int64_t i, char* name, size_t name_len, double* vals7);
int kx_set_act_prescale(kx_model* m, const char* conv_name, int log2_scale);

/* ---- voice table on device + output packing -----------------------------
 * kx_set_voice_table uploads the table that `TTSKoko::load_voices` builds (kokorox/src/tts/koko.rs:1308-1334;
 * layout [n_voices][511][1][256] f32, row 510 zero, kokorox/src/utils/hf_cache.rs:284-309) once; requests then
 * carry (voice id, weight) pairs instead of 256 floats.  The mix is `mix_styles` (koko.rs:1255-1306) on the
 * GPU, bit for bit: max_mix = 1 copies row `lens[b] - 2` of the voice; otherwise the row is
 * sum_k row_k * (weights[k] * 0.1), in order, un-normalised; entries with voice id < 0 are skipped. */
int kx_set_voice_table(kx_model* m, const float* table, int n_voices);

/* Output forms of the reference: 0 = f32 mono (ort_koko.rs:80-87), 1 = f32 stereo with every sample written
 * twice (koko.rs:1239-1246), 2 = 16-bit PCM `(s.clamp(-1,1) * 32767) as i16` (kokorox-websocket/src/lib.rs:701-704).
 * Packed on the GPU, so only the packed bytes cross PCIe. */
#define KX_PACK_F32_MONO 0
#define KX_PACK_F32_STEREO 1
#define KX_PACK_PCM16_MONO 2

/* kx_infer with device-side style lookup/mix and packed output.  *out = library-owned bytes of the B utterances back
 * to back (kx_free_packed); out_bytes[b] / out_samples[b] per utterance. */
int kx_infer_voices(kx_model* m, const int64_t* ids, int64_t t_stride, const int32_t* lens, int B,
                    const int32_t* voice_ids, const float* weights, int max_mix, const float* speeds, int n_speed,
                    uint64_t seed, uint32_t flags, int format, void** out, int64_t* out_bytes, int64_t* out_samples);
/* kx_infer (explicit style rows) with packed output. */
int kx_infer_packed(kx_model* m, const int64_t* ids, int64_t t_stride, const int32_t* lens, int B, const float* styles,
                    const float* speeds, int n_speed, uint64_t seed, uint32_t flags, int format, void** out,
                    int64_t* out_bytes, int64_t* out_samples);
void kx_free_packed(void* p);

/* ---- request dispatcher ----------------------------------------------------
 * Replaces the reference's one-request-at-a-time `Mutex<Session>` (kokorox/src/onn/ort_koko.rs:78; callers
 * kokorox-openai/src/lib.rs:370-439, kokorox-websocket/src/lib.rs:657-668).  Any number of threads submit
 * single utterances; two workers per model (= per GPU; one runs a forward while the other hands the previous batch's results
 * to its clients) coalesce whatever is queued — up to max_batch, waiting at most max_wait_us after the first arrival — into
 * one batched forward.  A request's waveform depends only on (ids, style, speed, seed), not on what it was batched with, nor
 * on which model ran it, nor on which LSTM recurrence that model was running (kx_model_status: both forms give the same bits).
 * A request is checked completely when it is submitted (token ids against the models' embedding tables, token count, speed,
 * format, voice ids against the smallest voice table among the models, mix size) and refused there with KX_ERR_INVALID: a bad
 * request never reaches a batch.  If a batch still fails as a whole, an INVALID-class failure is re-run request by request
 * (only the requests that fail alone report it); a DEVICE-class failure is retried once as a batch, and if it fails again
 * that MODEL is marked failed: it takes no more work and its batch goes back to the head of the queue for the healthy models
 * (kx_dispatcher_health).  Requests report KX_ERR_DEVICE only when no healthy model is left.
 * Results (*out of the submit calls) are NOT malloc'd memory: they point into a page-locked buffer shared by the requests of
 * one batch, which returns to the library's pool when the last of them has been released.  Release them with kx_free_audio /
 * kx_free_packed and with nothing else (free() corrupts the heap); a result that is kept alive keeps its batch's buffer alive. */
typedef struct kx_dispatcher kx_dispatcher;
kx_dispatcher* kx_dispatcher_create(kx_model** models, int n_models, int max_batch, int max_wait_us, char* err,
                                    size_t err_len);
/* The same, after one discarded forward of max_batch utterances x warm_tokens tokens at warm_frames_per_token frames per token
 * on every model (kx_warmup, the models side by side): what a server should call, so that no request ever pays for an arena
 * growing (a stream sync + hipFree + hipMalloc of gigabytes: a 1.3 s outlier in the round-4 soak). */
kx_dispatcher* kx_dispatcher_create_warm(kx_model** models, int n_models, int max_batch, int max_wait_us, int warm_tokens,
                                         int warm_frames_per_token, char* err, size_t err_len);
/* Blocking; thread-safe.  ids = n_tokens ids incl. the two 0 pads; style = 256 floats.  *out: release with kx_free_audio
 * only (see above).  Equals kx_infer(B = 1, same seed, utterance base 0) bit for bit. */
int kx_dispatcher_submit(kx_dispatcher* d, const int64_t* ids, int n_tokens, const float* style, float speed,
                         uint64_t seed, float** out, int64_t* out_len, char* err, size_t err_len);
/* The request as the reference's servers make it (kokorox-openai/src/lib.rs:370-439, kokorox-websocket/src/lib.rs:
 * 657-736): the voice is EITHER the 256-float style row (`style`, voice_ids = NULL) OR names into the device voice
 * table every model of the dispatcher holds (kx_set_voice_table): `voice_ids[n_mix]` with `weights` = NULL and
 * n_mix = 1 for a single voice (row copy), or with `weights[n_mix]` for a mix "a.4+b.5" (sum_k row_k * (w_k * 0.1),
 * koko.rs:1255-1306; ids < 0 are skipped); `format` is a KX_PACK_* output form.  *out = the packed bytes (release with
 * kx_free_audio / kx_free_packed only); requests of every kind and format share batches, and a request's bytes equal those of
 * kx_infer_voices / kx_infer_packed (B = 1, same seed, utterance base 0) whatever it was batched with.
 * Errors are per request: when a batch fails as a whole its requests are re-run one by one. */
int kx_dispatcher_submit_ex(kx_dispatcher* d, const int64_t* ids, int n_tokens, const float* style,
                            const int32_t* voice_ids, const float* weights, int n_mix, float speed, uint64_t seed,
                            int format, void** out, int64_t* out_bytes, int64_t* out_samples, char* err, size_t err_len);
int kx_dispatcher_stats(kx_dispatcher* d, int64_t* n_requests, int64_t* n_batches, int64_t* max_batch_seen);
/* batches each model (worker) has run so far: per_model[n_models] */
int kx_dispatcher_model_batches(kx_dispatcher* d, int64_t* per_model, int n_models);
/* What went wrong so far: requests that were re-run one by one after their batch failed with an INVALID-class error, and
 * batches that were run a second time after a DEVICE-class failure.  Both stay 0 in a healthy deployment. */
int kx_dispatcher_failures(kx_dispatcher* d, int64_t* n_replayed, int64_t* n_retried);
/* Per-model health: healthy[i] = 1 while model i takes work, 0 once a batch failed on it with KX_ERR_DEVICE twice in a row (it
 * is then out for the dispatcher's lifetime: destroy the dispatcher and the model to recover the GPU); *n_model_failures =
 * models marked failed so far, *n_requeued = requests that went back to the queue for another model (either may be NULL). */
int kx_dispatcher_health(kx_dispatcher* d, int32_t* healthy, int n_models, int64_t* n_model_failures, int64_t* n_requeued);
void kx_dispatcher_destroy(kx_dispatcher* d);  /* waits for queued requests; models stay alive */

/* ---- debugging ---------------------------------------------------------------------
 * (the stand-alone kernel hooks of tests/ are NOT part of this library: include/kokorox_hip_test.h, libkokorox_hip_test.so) */

/* Named intermediate of the last kx_infer*(…, KX_FLAG_TAPS): utterance b's [C, L] slab
 * is copied to out (row-major, rows = channels).  Query with out = NULL to get C and L. */
int kx_debug_tap(kx_model* m, const char* name, int b, float* out, int64_t out_cap,
                 int32_t* C, int32_t* L);

const char* kx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KOKOROX_HIP_H */
