//! `kokorox-hip`: drop-in for `kokorox::onn::ort_koko::OrtKoko` on AMD MI355X (gfx950).
//!
//! The reference runs the Kokoro-82M graph through ONNX Runtime (`ort::Session::run`,
//! kokorox/src/onn/ort_koko.rs:79).  This crate binds `libkokorox_hip.so` (include/kokorox_hip.h), whose forward
//! pass is hand-written CDNA4 HIP, and exposes the same two public operations with the same signatures:
//!
//! ```text
//! OrtKoko::new(model_path: String) -> Result<Self, String>                               ort_koko.rs:31-35
//! OrtKoko::infer(&self, tokens: Vec<Vec<i64>>, styles: Vec<Vec<f32>>, speed: f32)
//!     -> Result<ArrayBase<OwnedRepr<f32>, IxDyn>, Box<dyn Error>>                        ort_koko.rs:37-91
//! ```
//!
//! Change in the reference: `use kokorox_hip::HipKoko as OrtKoko;` at kokorox/src/tts/koko.rs:40,570-573 and
//! `kokorox_hip::init(0)` in place of `init_ort` (koko/src/main.rs:1429).  `model_path` stays what the reference
//! passes: the Hugging Face `onnx/model.onnx` (kokorox/src/utils/hf_cache.rs:128-158; the fp16 / int8 / 4-bit
//! variants of hf_cache.rs:135-144 load too, de-quantised).  The library reads the ONNX initialisers itself (no ONNX
//! Runtime, no protobuf crate); a pre-converted `KXHIPW01` container (`import_onnx` below, or
//! `python -m kokorox_amd.importer`) is recognised by its magic and skips the conversion.
#![allow(clippy::too_many_arguments)]

use ndarray::{ArrayBase, IxDyn, OwnedRepr};
use std::error::Error;
use std::ffi::{c_char, c_int, c_void, CStr, CString};
use std::ptr;

pub const KX_OK: c_int = 0;
pub const KX_STYLE_DIM: usize = 256;
pub const KX_SAMPLES_PER_FRAME: usize = 600;
pub const KX_FLAG_NOISE_OFF: u32 = 1;
pub const KX_PACK_F32_MONO: c_int = 0;
pub const KX_PACK_F32_STEREO: c_int = 1;
pub const KX_PACK_PCM16_MONO: c_int = 2;

#[repr(C)]
pub struct KxModel {
    _p: [u8; 0],
}
#[repr(C)]
pub struct KxDispatcher {
    _p: [u8; 0],
}

// One declaration per entry point of include/kokorox_hip.h (test hooks excluded), same order, same arity.
extern "C" {
    fn kx_init(device_id: c_int, err: *mut c_char, err_len: usize) -> c_int;
    fn kx_create(weights_path: *const c_char, device_id: c_int, err: *mut c_char, err_len: usize) -> *mut KxModel;
    fn kx_import_onnx(onnx_path: *const c_char, out_path: *const c_char, err: *mut c_char, err_len: usize) -> c_int;
    fn kx_create_from_device_blob(d_blob: *const c_void, n_bytes: usize, device_id: c_int, err: *mut c_char,
                                  err_len: usize) -> *mut KxModel;
    fn kx_create_replicas(weights_path: *const c_char, device_ids: *const c_int, n: c_int,
                          out_models: *mut *mut KxModel, err: *mut c_char, err_len: usize) -> c_int;
    fn kx_replicas_times(out3: *mut f64) -> c_int;
    fn kx_create_partition(weights_path: *const c_char, device_id: c_int, part: c_int, n_parts: c_int,
                           err: *mut c_char, err_len: usize) -> *mut KxModel;
    fn kx_destroy(m: *mut KxModel);
    fn kx_last_error(m: *const KxModel) -> *const c_char;
    fn kx_last_error_copy(m: *const KxModel, buf: *mut c_char, buf_len: usize) -> c_int;
    fn kx_infer(m: *mut KxModel, ids: *const i64, t_stride: i64, lens: *const i32, b: c_int, styles: *const f32,
                speeds: *const f32, n_speed: c_int, seed: u64, flags: u32, out: *mut *mut f32,
                out_lens: *mut i64) -> c_int;
    fn kx_free_audio(p: *mut f32);
    fn kx_infer_device(m: *mut KxModel, d_ids: *const i64, t_stride: i64, lens_host: *const i32, b: c_int,
                       d_styles: *const f32, speeds_host: *const f32, n_speed: c_int, seed: u64, flags: u32,
                       d_audio: *mut f32, audio_ld: i64, d_frames: *mut i32, need_ld: *mut i64) -> c_int;
    fn kx_sync(m: *mut KxModel) -> c_int;
    fn kx_warmup(m: *mut KxModel, b: c_int, n_tokens: c_int, frames_per_token: c_int) -> c_int;
    fn kx_arena_bytes(m: *mut KxModel, out3: *mut i64) -> c_int;
    fn kx_call_times(m: *mut KxModel, out4: *mut f64) -> c_int;
    fn kx_model_status(m: *mut KxModel, out4: *mut i64) -> c_int;
    fn kx_model_info(m: *mut KxModel, out8: *mut i64) -> c_int;
    fn kx_set_pinned_durations(m: *mut KxModel, pattern: *const i32, n: c_int) -> c_int;
    fn kx_set_conv_mode(m: *mut KxModel, mode: c_int) -> c_int;
    fn kx_get_conv_mode(m: *mut KxModel) -> c_int;
    fn kx_set_stft_variant(m: *mut KxModel, variant: c_int) -> c_int;
    fn kx_get_stft_variant(m: *mut KxModel) -> c_int;
    fn kx_set_utterance_base(m: *mut KxModel, utt_base: u64) -> c_int;
    fn kx_set_lanes(m: *mut KxModel, n_lanes: c_int) -> c_int;
    fn kx_profile_enable(m: *mut KxModel, on: c_int) -> c_int;
    fn kx_profile_read(m: *mut KxModel, launches: *mut i64, total_ms: *mut f64, total_flops: *mut f64) -> c_int;
    fn kx_profile_detail(m: *mut KxModel, out: *mut f64, cap_rows: i64, n_rows: *mut i64) -> c_int;
    fn kx_profile_aux(m: *mut KxModel, stats_launches: *mut i64, stats_bytes: *mut f64) -> c_int;
    fn kx_diag_enable(m: *mut KxModel, on: c_int) -> c_int;
    fn kx_diag_count(m: *mut KxModel, n: *mut i64) -> c_int;
    fn kx_diag_get(m: *mut KxModel, i: i64, name: *mut c_char, name_len: usize, vals7: *mut f64) -> c_int;
    fn kx_set_act_prescale(m: *mut KxModel, conv_name: *const c_char, log2_scale: c_int) -> c_int;
    fn kx_set_voice_table(m: *mut KxModel, table: *const f32, n_voices: c_int) -> c_int;
    fn kx_infer_voices(m: *mut KxModel, ids: *const i64, t_stride: i64, lens: *const i32, b: c_int,
                       voice_ids: *const i32, weights: *const f32, max_mix: c_int, speeds: *const f32,
                       n_speed: c_int, seed: u64, flags: u32, format: c_int, out: *mut *mut c_void,
                       out_bytes: *mut i64, out_samples: *mut i64) -> c_int;
    fn kx_infer_packed(m: *mut KxModel, ids: *const i64, t_stride: i64, lens: *const i32, b: c_int,
                       styles: *const f32, speeds: *const f32, n_speed: c_int, seed: u64, flags: u32,
                       format: c_int, out: *mut *mut c_void, out_bytes: *mut i64, out_samples: *mut i64) -> c_int;
    fn kx_free_packed(p: *mut c_void);
    fn kx_dispatcher_create(models: *mut *mut KxModel, n_models: c_int, max_batch: c_int, max_wait_us: c_int,
                            err: *mut c_char, err_len: usize) -> *mut KxDispatcher;
    fn kx_dispatcher_create_warm(models: *mut *mut KxModel, n_models: c_int, max_batch: c_int, max_wait_us: c_int,
                                 warm_tokens: c_int, warm_frames_per_token: c_int, err: *mut c_char, err_len: usize) -> *mut KxDispatcher;
    fn kx_dispatcher_submit(d: *mut KxDispatcher, ids: *const i64, n_tokens: c_int, style: *const f32, speed: f32,
                            seed: u64, out: *mut *mut f32, out_len: *mut i64, err: *mut c_char,
                            err_len: usize) -> c_int;
    fn kx_dispatcher_submit_ex(d: *mut KxDispatcher, ids: *const i64, n_tokens: c_int, style: *const f32,
                               voice_ids: *const i32, weights: *const f32, n_mix: c_int, speed: f32, seed: u64,
                               format: c_int, out: *mut *mut c_void, out_bytes: *mut i64, out_samples: *mut i64,
                               err: *mut c_char, err_len: usize) -> c_int;
    fn kx_dispatcher_stats(d: *mut KxDispatcher, n_requests: *mut i64, n_batches: *mut i64,
                           max_batch_seen: *mut i64) -> c_int;
    fn kx_dispatcher_failures(d: *mut KxDispatcher, n_replayed: *mut i64, n_retried: *mut i64) -> c_int;
    fn kx_dispatcher_model_batches(d: *mut KxDispatcher, per_model: *mut i64, n_models: c_int) -> c_int;
    fn kx_dispatcher_health(d: *mut KxDispatcher, healthy: *mut i32, n_models: c_int, n_model_failures: *mut i64,
                            n_requeued: *mut i64) -> c_int;
    fn kx_dispatcher_destroy(d: *mut KxDispatcher);
    fn kx_version() -> *const c_char;
}

fn cstr_buf(buf: &[c_char]) -> String {
    unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned()
}

/// Replaces `init_ort(dylib_path)` (kokorox/src/onn/mod.rs:19-49): checks that `device_id` is a gfx950 part.
pub fn init(device_id: i32) -> Result<(), String> {
    let mut err = vec![0 as c_char; 512];
    match unsafe { kx_init(device_id, err.as_mut_ptr(), err.len()) } {
        KX_OK => Ok(()),
        _ => Err(cstr_buf(&err)),
    }
}

/// Host only: `model.onnx` -> the KXHIPW01 container `HipKoko::new` would build from it in memory (convert once, load
/// faster afterwards).  No reference counterpart: ONNX Runtime parses the file on every start (ort_base.rs:27-33).
pub fn import_onnx(onnx_path: &str, out_path: &str) -> Result<(), String> {
    let a = CString::new(onnx_path).map_err(|e| e.to_string())?;
    let b = CString::new(out_path).map_err(|e| e.to_string())?;
    let mut err = vec![0 as c_char; 1024];
    match unsafe { kx_import_onnx(a.as_ptr(), b.as_ptr(), err.as_mut_ptr(), err.len()) } {
        KX_OK => Ok(()),
        _ => Err(cstr_buf(&err)),
    }
}

pub fn version() -> String {
    unsafe { CStr::from_ptr(kx_version()) }.to_string_lossy().into_owned()
}

/// Drop-in for `kokorox::onn::ort_koko::OrtKoko` (same two public methods, same signatures).
pub struct HipKoko {
    h: *mut KxModel,
}
// Calls on one model are serialised inside the library, exactly as `Mutex<Session>` does (ort_koko.rs:14,17-18,78).
unsafe impl Send for HipKoko {}
unsafe impl Sync for HipKoko {}

impl HipKoko {
    /// `OrtKoko::new` (ort_koko.rs:31-35): load failure is an `Err(String)`; the caller `.expect`s (koko.rs:572).
    /// `model_path` is the `.onnx` the reference passes (koko.rs:570-573) or a pre-converted KXHIPW01 container.
    pub fn new(model_path: String) -> Result<Self, String> {
        Self::on_device(model_path, 0)
    }

    pub fn on_device(model_path: String, device_id: i32) -> Result<Self, String> {
        let p = CString::new(model_path).map_err(|e| e.to_string())?;
        let mut err = vec![0 as c_char; 512];
        let h = unsafe { kx_create(p.as_ptr(), device_id, err.as_mut_ptr(), err.len()) };
        if h.is_null() {
            return Err(cstr_buf(&err));
        }
        Ok(HipKoko { h })
    }

    /// One model per device id from ONE read of the weight file (device-to-device fan-out over xGMI inside the
    /// library).  The result is what `HipKokoDispatcher::new` takes; the reference's server is one process
    /// (kokorox-openai/src/lib.rs:370-439).
    pub fn replicas(model_path: String, device_ids: &[i32]) -> Result<Vec<Self>, String> {
        let p = CString::new(model_path).map_err(|e| e.to_string())?;
        let mut err = vec![0 as c_char; 512];
        let mut hs: Vec<*mut KxModel> = vec![ptr::null_mut(); device_ids.len()];
        let rc = unsafe {
            kx_create_replicas(p.as_ptr(), device_ids.as_ptr(), device_ids.len() as c_int, hs.as_mut_ptr(),
                               err.as_mut_ptr(), err.len())
        };
        if rc != KX_OK {
            return Err(cstr_buf(&err));
        }
        Ok(hs.into_iter().map(|h| HipKoko { h }).collect())
    }

    /// A model confined to CUs `[part, part + 1) x CUs / n_parts` of the device: partitions run their forwards side by side on one
    /// GPU without competing for a CU (`kx_create_replicas` does this for a device id that is named several times).
    pub fn partition(model_path: String, device_id: i32, part: i32, n_parts: i32) -> Result<Self, String> {
        let p = CString::new(model_path).map_err(|e| e.to_string())?;
        let mut err = vec![0 as c_char; 512];
        let h = unsafe { kx_create_partition(p.as_ptr(), device_id, part, n_parts, err.as_mut_ptr(), err.len()) };
        if h.is_null() {
            return Err(cstr_buf(&err));
        }
        Ok(HipKoko { h })
    }

    /// Milestones of the process's last `replicas` call in ms from its entry: file read (+ conversion), blob resident on every
    /// device, every model built.
    pub fn replicas_times() -> [f64; 3] {
        let mut t = [0f64; 3];
        unsafe { kx_replicas_times(t.as_mut_ptr()) };
        t
    }

    /// `[weight-file variant, reference arithmetic (0 for the de-quantised int8 / q4 variants), conv mode, vocabulary rows,
    /// voices, CU partition, partitions, CUs]` (kx_model_info).
    pub fn info(&self) -> Result<[i64; 8], Box<dyn Error>> {
        let mut o = [0i64; 8];
        self.check(unsafe { kx_model_info(self.h, o.as_mut_ptr()) })?;
        Ok(o)
    }

    /// `[recurrence form (0 resident weights / 1 streaming fall-back), hand-off time-outs, clean forwards until the resident forms
    /// return, transparently re-run calls]`: all zero in a healthy deployment (kx_model_status).
    pub fn status(&self) -> Result<[i64; 4], Box<dyn Error>> {
        let mut o = [0i64; 4];
        self.check(unsafe { kx_model_status(self.h, o.as_mut_ptr()) })?;
        Ok(o)
    }

    /// Host-side milestones of the last call in ms from its entry (kx_call_times).
    pub fn call_times(&self) -> Result<[f64; 4], Box<dyn Error>> {
        let mut o = [0f64; 4];
        self.check(unsafe { kx_call_times(self.h, o.as_mut_ptr()) })?;
        Ok(o)
    }

    /// A model from a weight blob that already sits in this GPU's memory (after an RCCL broadcast).
    ///
    /// # Safety
    /// `d_blob` must be a device pointer valid for `n_bytes` on `device_id`.
    pub unsafe fn from_device_blob(d_blob: *const c_void, n_bytes: usize, device_id: i32) -> Result<Self, String> {
        let mut err = vec![0 as c_char; 512];
        let h = kx_create_from_device_blob(d_blob, n_bytes, device_id, err.as_mut_ptr(), err.len());
        if h.is_null() {
            return Err(cstr_buf(&err));
        }
        Ok(HipKoko { h })
    }

    fn last_error(&self) -> String {
        let mut buf = vec![0 as c_char; 512];
        unsafe { kx_last_error_copy(self.h, buf.as_mut_ptr(), buf.len()) };
        let s = cstr_buf(&buf);
        if s.is_empty() {
            // (same text through the pointer form; kept so that both entry points stay bound)
            unsafe { CStr::from_ptr(kx_last_error(self.h)) }.to_string_lossy().into_owned()
        } else {
            s
        }
    }

    fn check(&self, rc: c_int) -> Result<(), Box<dyn Error>> {
        if rc == KX_OK {
            Ok(())
        } else {
            Err(format!("kokorox_hip error {}: {}", rc, self.last_error()).into())
        }
    }

    fn flatten(tokens: &[Vec<i64>]) -> (Vec<i64>, Vec<i32>, usize) {
        let stride = tokens.iter().map(|t| t.len()).max().unwrap_or(0).max(1);
        let lens: Vec<i32> = tokens.iter().map(|t| t.len() as i32).collect();
        let mut ids = vec![0i64; tokens.len() * stride];
        for (i, t) in tokens.iter().enumerate() {
            ids[i * stride..i * stride + t.len()].copy_from_slice(t);
        }
        (ids, lens, stride)
    }

    /// `OrtKoko::infer` (ort_koko.rs:37-91).  B = 1 behaves exactly like the reference call at koko.rs:1177;
    /// B > 1 returns the B waveforms back to back (the chunk loop of koko.rs:947-1191 as one call).
    /// An empty token list is an `Err`, not the index panic of ort_koko.rs:56.
    pub fn infer(&self, tokens: Vec<Vec<i64>>, styles: Vec<Vec<f32>>, speed: f32)
        -> Result<ArrayBase<OwnedRepr<f32>, IxDyn>, Box<dyn Error>> {
        let (wav, _lens) = self.infer_batch(&tokens, &styles, &[speed], 0, 0)?;
        let n = wav.len();
        Ok(ArrayBase::from_shape_vec(IxDyn(&[n]), wav)?)
    }

    /// Batched form with an explicit noise seed; returns (samples back to back, per-utterance sample counts).
    pub fn infer_batch(&self, tokens: &[Vec<i64>], styles: &[Vec<f32>], speeds: &[f32], seed: u64, flags: u32)
        -> Result<(Vec<f32>, Vec<i64>), Box<dyn Error>> {
        if tokens.is_empty() || tokens[0].is_empty() {
            return Err("infer: empty token list".into());
        }
        if styles.len() != tokens.len() || styles.iter().any(|s| s.len() != KX_STYLE_DIM) {
            return Err("infer: one style row of 256 floats per utterance is required".into());
        }
        let b = tokens.len();
        let (ids, lens, stride) = Self::flatten(tokens);
        let st: Vec<f32> = styles.iter().flatten().copied().collect();
        let mut out: *mut f32 = ptr::null_mut();
        let mut out_lens = vec![0i64; b];
        let rc = unsafe {
            kx_infer(self.h, ids.as_ptr(), stride as i64, lens.as_ptr(), b as c_int, st.as_ptr(), speeds.as_ptr(),
                     speeds.len() as c_int, seed, flags, &mut out, out_lens.as_mut_ptr())
        };
        self.check(rc)?;
        let n: usize = out_lens.iter().sum::<i64>() as usize;
        let v = unsafe { std::slice::from_raw_parts(out, n) }.to_vec(); // the same owned copy as ort_koko.rs:85
        unsafe { kx_free_audio(out) };
        Ok((v, out_lens))
    }

    /// Device-resident voice table (`TTSKoko::load_voices` layout, koko.rs:1308-1334): `[n_voices][511][256]`.
    pub fn set_voice_table(&self, table: &[f32], n_voices: usize) -> Result<(), Box<dyn Error>> {
        if table.len() != n_voices * 511 * KX_STYLE_DIM {
            return Err("voice table must hold n_voices * 511 * 256 floats".into());
        }
        self.check(unsafe { kx_set_voice_table(self.h, table.as_ptr(), n_voices as c_int) })
    }

    /// `mix_styles` on the GPU (koko.rs:1255-1306): `(voice id, weight)` pairs per utterance, packed output bytes.
    pub fn infer_voices(&self, tokens: &[Vec<i64>], voice_ids: &[i32], weights: &[f32], max_mix: usize,
                        speeds: &[f32], seed: u64, format: c_int) -> Result<(Vec<u8>, Vec<i64>), Box<dyn Error>> {
        if tokens.is_empty() || voice_ids.len() != tokens.len() * max_mix || weights.len() != voice_ids.len() {
            return Err("infer_voices: B * max_mix voice ids and weights are required".into());
        }
        let b = tokens.len();
        let (ids, lens, stride) = Self::flatten(tokens);
        let mut out: *mut c_void = ptr::null_mut();
        let (mut nbytes, mut nsamp) = (vec![0i64; b], vec![0i64; b]);
        let rc = unsafe {
            kx_infer_voices(self.h, ids.as_ptr(), stride as i64, lens.as_ptr(), b as c_int, voice_ids.as_ptr(),
                            weights.as_ptr(), max_mix as c_int, speeds.as_ptr(), speeds.len() as c_int, seed, 0,
                            format, &mut out, nbytes.as_mut_ptr(), nsamp.as_mut_ptr())
        };
        self.check(rc)?;
        let n: usize = nbytes.iter().sum::<i64>() as usize;
        let v = unsafe { std::slice::from_raw_parts(out as *const u8, n) }.to_vec();
        unsafe { kx_free_packed(out) };
        Ok((v, nbytes))
    }

    /// Explicit style rows, output packed on the GPU: f32 stereo (koko.rs:1239-1246) or PCM16
    /// (kokorox-websocket/src/lib.rs:701-704).
    pub fn infer_packed(&self, tokens: &[Vec<i64>], styles: &[Vec<f32>], speeds: &[f32], seed: u64, format: c_int)
        -> Result<(Vec<u8>, Vec<i64>), Box<dyn Error>> {
        if tokens.is_empty() || styles.len() != tokens.len() {
            return Err("infer_packed: one style row per utterance is required".into());
        }
        let b = tokens.len();
        let (ids, lens, stride) = Self::flatten(tokens);
        let st: Vec<f32> = styles.iter().flatten().copied().collect();
        let mut out: *mut c_void = ptr::null_mut();
        let (mut nbytes, mut nsamp) = (vec![0i64; b], vec![0i64; b]);
        let rc = unsafe {
            kx_infer_packed(self.h, ids.as_ptr(), stride as i64, lens.as_ptr(), b as c_int, st.as_ptr(),
                            speeds.as_ptr(), speeds.len() as c_int, seed, 0, format, &mut out, nbytes.as_mut_ptr(),
                            nsamp.as_mut_ptr())
        };
        self.check(rc)?;
        let n: usize = nbytes.iter().sum::<i64>() as usize;
        let v = unsafe { std::slice::from_raw_parts(out as *const u8, n) }.to_vec();
        unsafe { kx_free_packed(out) };
        Ok((v, nbytes))
    }

    /// Raw device-pointer form (inputs and output stay in HBM).
    ///
    /// # Safety
    /// Every pointer must be device memory of this model's GPU with the extents documented in kokorox_hip.h.
    pub unsafe fn infer_device(&self, d_ids: *const i64, t_stride: i64, lens_host: &[i32], d_styles: *const f32,
                               speeds_host: &[f32], seed: u64, flags: u32, d_audio: *mut f32, audio_ld: i64,
                               d_frames: *mut i32) -> Result<i64, Box<dyn Error>> {
        let mut need: i64 = 0;
        let rc = kx_infer_device(self.h, d_ids, t_stride, lens_host.as_ptr(), lens_host.len() as c_int, d_styles,
                                 speeds_host.as_ptr(), speeds_host.len() as c_int, seed, flags, d_audio, audio_ld,
                                 d_frames, &mut need);
        self.check(rc)?;
        Ok(need)
    }

    pub fn sync(&self) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { kx_sync(self.h) })
    }
    pub fn set_pinned_durations(&self, pattern: &[i32]) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { kx_set_pinned_durations(self.h, pattern.as_ptr(), pattern.len() as c_int) })
    }
    /// For servers: one discarded forward at the largest (batch x length) shape the deployment expects, so that the arenas and
    /// the page-locked result buffer exist before the first request (an arena that grows under load is a second-long outlier).
    pub fn warmup(&self, batch: i32, n_tokens: i32, frames_per_token: i32) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { kx_warmup(self.h, batch, n_tokens, frames_per_token) })
    }
    /// Capacities in bytes of the token-axis, frame-axis and I/O arenas.
    pub fn arena_bytes(&self) -> Result<[i64; 3], Box<dyn Error>> {
        let mut v = [0i64; 3];
        self.check(unsafe { kx_arena_bytes(self.h, v.as_mut_ptr()) })?;
        Ok(v)
    }
    /// Contraction arithmetic: 6 = f16f8 (the default: split f16 products, the cross terms of the 7- / 11-tap convs on 8-bit MFMAs),
    /// 1 = f16x3 (three f16 MFMAs per product), 0 = f32 MFMA, 4 / 5 = one f16 / bf16 MFMA per product (opt-in reduced precision).
    pub fn set_conv_mode(&self, mode: i32) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { kx_set_conv_mode(self.h, mode) })
    }
    pub fn conv_mode(&self) -> i32 {
        unsafe { kx_get_conv_mode(self.h) }
    }
    /// 0 = the ONNX export's conv-based STFT pair (upstream `CustomSTFT`), 1 = `torch.stft/istft` semantics.
    pub fn set_stft_variant(&self, variant: i32) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { kx_set_stft_variant(self.h, variant) })
    }
    pub fn stft_variant(&self) -> i32 {
        unsafe { kx_get_stft_variant(self.h) }
    }
    /// 1 = one stream; 4 = the independent chains of the back half on streams of their own; 0 (default) = by batch size.
    pub fn set_lanes(&self, n: i32) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { kx_set_lanes(self.h, n) })
    }
    pub fn set_utterance_base(&self, base: u64) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { kx_set_utterance_base(self.h, base) })
    }
    /// Real-weights diagnostics: measure every conv input during the following calls.
    pub fn diag_enable(&self, on: bool) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { kx_diag_enable(self.h, on as c_int) })
    }
    /// (layer, [rows, Cin, taps, pre-scale exponent, absmax, rms, elements]) per conv launch since `diag_enable(true)`.
    pub fn diag_records(&self) -> Result<Vec<(String, [f64; 7])>, Box<dyn Error>> {
        let mut n = 0i64;
        self.check(unsafe { kx_diag_count(self.h, &mut n) })?;
        let mut out = Vec::with_capacity(n as usize);
        for i in 0..n {
            let mut name = vec![0 as c_char; 128];
            let mut v = [0f64; 7];
            self.check(unsafe { kx_diag_get(self.h, i, name.as_mut_ptr(), name.len(), v.as_mut_ptr()) })?;
            out.push((cstr_buf(&name), v));
        }
        Ok(out)
    }
    /// f16x3: multiply a layer's transformed input by 2^log2_scale before the hi/lo split (undone exactly).
    pub fn set_act_prescale(&self, layer: &str, log2_scale: i32) -> Result<(), Box<dyn Error>> {
        let c = CString::new(layer)?;
        self.check(unsafe { kx_set_act_prescale(self.h, c.as_ptr(), log2_scale) })
    }
    pub fn profile_enable(&self, on: bool) -> Result<(), Box<dyn Error>> {
        self.check(unsafe { kx_profile_enable(self.h, on as c_int) })
    }
    /// (launches, milliseconds, algorithmic FLOPs) of the dominant kernel family since the last read.
    pub fn profile_read(&self) -> Result<(i64, f64, f64), Box<dyn Error>> {
        let (mut n, mut ms, mut fl) = (0i64, 0f64, 0f64);
        self.check(unsafe { kx_profile_read(self.h, &mut n, &mut ms, &mut fl) })?;
        Ok((n, ms, fl))
    }
    /// (launches, bytes) of the unfused InstanceNorm statistics passes since the last call.
    pub fn profile_aux(&self) -> Result<(i64, f64), Box<dyn Error>> {
        let (mut n, mut by) = (0i64, 0f64);
        self.check(unsafe { kx_profile_aux(self.h, &mut n, &mut by) })?;
        Ok((n, by))
    }
    /// Per-launch rows of the last `profile_read`: [rows, Cin, taps, dil, stride, store, cols, flops, ms, bytes].
    pub fn profile_detail(&self) -> Result<Vec<[f64; 10]>, Box<dyn Error>> {
        let mut n = 0i64;
        self.check(unsafe { kx_profile_detail(self.h, ptr::null_mut(), 0, &mut n) })?;
        let mut rows = vec![[0f64; 10]; n as usize];
        self.check(unsafe { kx_profile_detail(self.h, rows.as_mut_ptr() as *mut f64, n, &mut n) })?;
        Ok(rows)
    }
}

impl Drop for HipKoko {
    fn drop(&mut self) {
        unsafe { kx_destroy(self.h) } // replaces TTSKoko::cleanup's sleep-and-drop (koko.rs:1338-1375)
    }
}

/// How a dispatched request names its voice.
pub enum Voice<'a> {
    Row(&'a [f32]),
    Single(i32),
    Mix(&'a [(i32, f32)]),
}

/// Batching front over one model per GPU: the replacement for the reference's one-request-at-a-time
/// `Mutex<Session>` (ort_koko.rs:78; kokorox-openai/src/lib.rs:370-439, kokorox-websocket/src/lib.rs:657-668).
/// `submit` blocks and may be called from any number of threads (tokio `spawn_blocking` in the servers).
pub struct HipKokoDispatcher {
    d: *mut KxDispatcher,
    _models: Vec<HipKoko>, // kept alive for the dispatcher's lifetime
}
unsafe impl Send for HipKokoDispatcher {}
unsafe impl Sync for HipKokoDispatcher {}

impl HipKokoDispatcher {
    pub fn new(models: Vec<HipKoko>, max_batch: i32, max_wait_us: i32) -> Result<Self, String> {
        let mut hs: Vec<*mut KxModel> = models.iter().map(|m| m.h).collect();
        let mut err = vec![0 as c_char; 256];
        let d = unsafe {
            kx_dispatcher_create(hs.as_mut_ptr(), hs.len() as c_int, max_batch, max_wait_us, err.as_mut_ptr(), err.len())
        };
        if d.is_null() {
            return Err(cstr_buf(&err));
        }
        Ok(HipKokoDispatcher { d, _models: models })
    }

    /// `new`, after one discarded forward of `max_batch` utterances x `warm_tokens` tokens at `warm_frames_per_token` frames per
    /// token on every model: what a server should call, so that no request ever pays for an arena growing.
    pub fn new_warm(models: Vec<HipKoko>, max_batch: i32, max_wait_us: i32, warm_tokens: i32, warm_frames_per_token: i32) -> Result<Self, String> {
        let mut hs: Vec<*mut KxModel> = models.iter().map(|m| m.h).collect();
        let mut err = vec![0 as c_char; 256];
        let d = unsafe {
            kx_dispatcher_create_warm(hs.as_mut_ptr(), hs.len() as c_int, max_batch, max_wait_us, warm_tokens, warm_frames_per_token,
                                      err.as_mut_ptr(), err.len())
        };
        if d.is_null() {
            return Err(cstr_buf(&err));
        }
        Ok(HipKokoDispatcher { d, _models: models })
    }

    /// Per-model health (true = takes work; false = failed a batch with a device error twice in a row and is out), the number of
    /// models marked failed and of requests that went back to the queue for another model.
    pub fn health(&self) -> (Vec<bool>, i64, i64) {
        let n = self._models.len();
        let mut h = vec![0i32; n];
        let (mut f, mut r) = (0i64, 0i64);
        unsafe { kx_dispatcher_health(self.d, h.as_mut_ptr(), n as c_int, &mut f, &mut r) };
        (h.into_iter().map(|v| v != 0).collect(), f, r)
    }

    /// One utterance: `ids` already wrapped with the two 0 pads (koko.rs:1169-1173), `style` = 256 floats.
    /// Bit-identical to `HipKoko::infer` of the same request alone, whatever it was batched with.
    pub fn submit(&self, ids: &[i64], style: &[f32], speed: f32, seed: u64) -> Result<Vec<f32>, Box<dyn Error>> {
        if style.len() != KX_STYLE_DIM {
            return Err("submit: the style row must have 256 floats".into());
        }
        let mut out: *mut f32 = ptr::null_mut();
        let mut n: i64 = 0;
        let mut err = vec![0 as c_char; 256];
        let rc = unsafe {
            kx_dispatcher_submit(self.d, ids.as_ptr(), ids.len() as c_int, style.as_ptr(), speed, seed, &mut out,
                                 &mut n, err.as_mut_ptr(), err.len())
        };
        if rc != KX_OK {
            return Err(format!("kokorox_hip error {}: {}", rc, cstr_buf(&err)).into());
        }
        let v = unsafe { std::slice::from_raw_parts(out, n as usize) }.to_vec();
        unsafe { kx_free_audio(out) };
        Ok(v)
    }

    /// The request as the servers make it (kokorox-openai/src/lib.rs:370-439): the voice by name into the device voice
    /// table -- `Voice::Single(id)` = row copy, `Voice::Mix(&[(id, weight)])` = "a.4+b.5" (koko.rs:1255-1306) -- or as
    /// the 256-float row, and the output form (0 f32 mono, 1 f32 stereo, 2 PCM16).  Returns the packed bytes and the
    /// sample count; bit-identical to the same request run alone.
    pub fn submit_ex(&self, ids: &[i64], voice: Voice, speed: f32, seed: u64, format: i32)
                     -> Result<(Vec<u8>, i64), Box<dyn Error>> {
        let (mut vid, mut w): (Vec<i32>, Vec<f32>) = (Vec::new(), Vec::new());
        let (style_p, vid_p, w_p, n_mix) = match voice {
            Voice::Row(s) => {
                if s.len() != KX_STYLE_DIM {
                    return Err("submit_ex: the style row must have 256 floats".into());
                }
                (s.as_ptr(), ptr::null(), ptr::null(), 0)
            }
            Voice::Single(id) => {
                vid.push(id);
                (ptr::null(), vid.as_ptr(), ptr::null(), 1)
            }
            Voice::Mix(parts) => {
                for (id, p) in parts {
                    vid.push(*id);
                    w.push(*p);
                }
                (ptr::null(), vid.as_ptr(), w.as_ptr(), vid.len() as c_int)
            }
        };
        let mut out: *mut c_void = ptr::null_mut();
        let (mut nb, mut ns) = (0i64, 0i64);
        let mut err = vec![0 as c_char; 256];
        let rc = unsafe {
            kx_dispatcher_submit_ex(self.d, ids.as_ptr(), ids.len() as c_int, style_p, vid_p, w_p, n_mix, speed, seed,
                                    format, &mut out, &mut nb, &mut ns, err.as_mut_ptr(), err.len())
        };
        if rc != KX_OK {
            return Err(format!("kokorox_hip error {}: {}", rc, cstr_buf(&err)).into());
        }
        let v = unsafe { std::slice::from_raw_parts(out as *const u8, nb as usize) }.to_vec();
        unsafe { kx_free_audio(out as *mut f32) };
        Ok((v, ns))
    }

    /// Batches each model (GPU) has run so far.
    pub fn model_batches(&self) -> Vec<i64> {
        let mut v = vec![0i64; self._models.len()];
        unsafe { kx_dispatcher_model_batches(self.d, v.as_mut_ptr(), v.len() as c_int) };
        v
    }

    /// (requests replayed one by one after an INVALID-class batch failure, batches retried after a DEVICE-class one).
    pub fn failures(&self) -> (i64, i64) {
        let (mut a, mut b) = (0i64, 0i64);
        unsafe { kx_dispatcher_failures(self.d, &mut a, &mut b) };
        (a, b)
    }

    /// (requests, batches, largest batch) so far.
    pub fn stats(&self) -> (i64, i64, i64) {
        let (mut a, mut b, mut c) = (0i64, 0i64, 0i64);
        unsafe { kx_dispatcher_stats(self.d, &mut a, &mut b, &mut c) };
        (a, b, c)
    }
}

impl Drop for HipKokoDispatcher {
    fn drop(&mut self) {
        unsafe { kx_dispatcher_destroy(self.d) } // waits for queued requests; the models drop afterwards
    }
}
