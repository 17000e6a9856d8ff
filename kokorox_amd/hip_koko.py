"""Host-side mirror of the reference's `OrtKoko` over the kokorox_hip C ABI (ctypes).

Reference interface being mirrored (/root/reference/kokorox/src/onn/ort_koko.rs):
    OrtKoko::new(model_path: String) -> Result<Self, String>                      (l.31-35)
    OrtKoko::infer(&self, tokens: Vec<Vec<i64>>, styles: Vec<Vec<f32>>, speed: f32)
        -> Result<ArrayBase<OwnedRepr<f32>, IxDyn>, Box<dyn Error>>               (l.37-91)
`HipKoko.new` / `HipKoko.infer` keep the names, argument meaning and error behaviour
(load failure and infer failure raise; an empty token list is an error instead of the
reference's index panic at l.56).  There is NO CPU fallback: if libkokorox_hip.so is
missing or no gfx950 device is visible, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libkokorox_hip.so")

PACK_F32_MONO, PACK_F32_STEREO, PACK_PCM16_MONO = 0, 1, 2
KX_OK, KX_ERR_INVALID, KX_ERR_IO, KX_ERR_DEVICE, KX_ERR_STATE = 0, 1, 2, 3, 4  # include/kokorox_hip.h
KX_FLAG_NOISE_OFF = 1
KX_FLAG_TAPS = 2
STYLE_DIM = 256
SAMPLES_PER_FRAME = 600

_lib = None


class KokoroxHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"kokorox_hip error {code}: {msg}")
        self.code = code


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libkokorox_hip.so (built by kokorox_amd.build / __graft_entry__.build())."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    import sys
    p = path or os.environ.get("KX_LIB") or LIB_PATH
    # torch wheels bundle their own libamdhip64; if this library (linked against /opt/rocm's) is loaded first
    # and torch later, the process ends up with two HIP runtimes and torch sees no GPU.  Loading torch first
    # makes both share one runtime.  Processes that never use torch (the C++ host) are unaffected.
    import sys
    if "torch" not in sys.modules and os.environ.get("KX_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(p) and p == LIB_PATH and os.environ.get("KX_NO_AUTOBUILD") != "1":
        # the prebuilt library normally travels with the tree; as a last resort build it here (hipcc, ~2-3 min)
        try:
            from . import build as _build
            _build.build_library(verbose=True)
        except Exception as e:  # fall through to the loud error below
            print(f"kokorox_amd: building libkokorox_hip.so failed: {e}", file=sys.stderr)
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} is missing: build it with `python -m kokorox_amd.build` (hipcc, gfx950). "
            "The HIP path has no CPU fallback.")
    lib = C.CDLL(p)
    vp, i32, i64, u32, u64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_float
    cp, sz = C.c_char_p, C.c_size_t
    sig = {
        "kx_version": (cp, []),
        "kx_init": (i32, [i32, cp, sz]),
        "kx_create": (vp, [cp, i32, cp, sz]),
        "kx_import_onnx": (i32, [cp, cp, cp, sz]),
        "kx_warmup": (i32, [vp, i32, i32, i32]),
        "kx_arena_bytes": (i32, [vp, C.POINTER(i64)]),
        "kx_call_times": (i32, [vp, C.POINTER(C.c_double)]),
        "kx_model_status": (i32, [vp, C.POINTER(i64)]),
        "kx_model_info": (i32, [vp, C.POINTER(i64)]),
        "kx_dispatcher_health": (i32, [vp, vp, i32, C.POINTER(i64), C.POINTER(i64)]),
        "kx_create_from_device_blob": (vp, [vp, sz, i32, cp, sz]),
        "kx_destroy": (None, [vp]),
        "kx_last_error": (cp, [vp]),
        "kx_last_error_copy": (i32, [vp, cp, sz]),
        "kx_create_replicas": (i32, [cp, vp, i32, vp, cp, sz]),
        "kx_replicas_times": (i32, [C.POINTER(C.c_double)]),
        "kx_create_partition": (vp, [cp, i32, i32, i32, cp, sz]),
        "kx_infer": (i32, [vp, vp, i64, vp, i32, vp, vp, i32, u64, u32, C.POINTER(C.POINTER(f32)), vp]),
        "kx_free_audio": (None, [C.POINTER(f32)]),
        "kx_infer_device": (i32, [vp, vp, i64, vp, i32, vp, vp, i32, u64, u32, vp, i64, vp, C.POINTER(i64)]),
        "kx_sync": (i32, [vp]),
        "kx_set_pinned_durations": (i32, [vp, vp, i32]),
        "kx_set_utterance_base": (i32, [vp, u64]),
        "kx_set_lanes": (i32, [vp, i32]),
        "kx_set_conv_mode": (i32, [vp, i32]),
        "kx_get_conv_mode": (i32, [vp]),
        "kx_set_stft_variant": (i32, [vp, i32]),
        "kx_get_stft_variant": (i32, [vp]),
        "kx_profile_enable": (i32, [vp, i32]),
        "kx_profile_read": (i32, [vp, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "kx_profile_detail": (i32, [vp, vp, i64, C.POINTER(i64)]),
        "kx_profile_aux": (i32, [vp, C.POINTER(i64), C.POINTER(C.c_double)]),
        "kx_diag_enable": (i32, [vp, i32]),
        "kx_diag_count": (i32, [vp, C.POINTER(i64)]),
        "kx_diag_get": (i32, [vp, i64, cp, sz, vp]),
        "kx_set_act_prescale": (i32, [vp, cp, i32]),
        "kx_set_voice_table": (i32, [vp, vp, i32]),
        "kx_infer_voices": (i32, [vp, vp, i64, vp, i32, vp, vp, i32, vp, i32, u64, u32, i32, C.POINTER(vp), vp, vp]),
        "kx_infer_packed": (i32, [vp, vp, i64, vp, i32, vp, vp, i32, u64, u32, i32, C.POINTER(vp), vp, vp]),
        "kx_free_packed": (None, [vp]),
        "kx_dispatcher_create": (vp, [vp, i32, i32, i32, cp, sz]),
        "kx_dispatcher_create_warm": (vp, [vp, i32, i32, i32, i32, i32, cp, sz]),
        "kx_dispatcher_submit": (i32, [vp, vp, i32, vp, f32, u64, C.POINTER(C.POINTER(f32)), C.POINTER(i64), cp, sz]),
        "kx_dispatcher_submit_ex": (i32, [vp, vp, i32, vp, vp, vp, i32, f32, u64, i32, C.POINTER(vp), C.POINTER(i64),
                                          C.POINTER(i64), cp, sz]),
        "kx_dispatcher_stats": (i32, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
        "kx_dispatcher_model_batches": (i32, [vp, vp, i32]),
        "kx_dispatcher_failures": (i32, [vp, C.POINTER(i64), C.POINTER(i64)]),
        "kx_dispatcher_destroy": (None, [vp]),
        "kx_debug_tap": (i32, [vp, cp, i32, vp, i64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


TEST_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libkokorox_hip_test.so")
_test_lib = None


def load_test_library() -> C.CDLL:
    """dlopen libkokorox_hip_test.so: the stand-alone kernel hooks of tests/ (include/kokorox_hip_test.h).  It links against
    libkokorox_hip.so, which is loaded first (same torch-first rule)."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    load_library()
    p = os.environ.get("KX_TEST_LIB") or TEST_LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} is missing: build it with `python -m kokorox_amd.build` (hipcc, gfx950)")
    lib = C.CDLL(p)
    vp, i32, i64, u32, u64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_float
    cp, sz = C.c_char_p, C.c_size_t
    sig = {
        "kx_test_conv1d": (i32, [i32, vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp,
                                 i32, i32, cp, sz]),
        "kx_test_conv1d_full": (i32, [i32, vp, i32, i32, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, i32, f32,
                                      f32, vp, vp, i32, cp, sz]),
        "kx_test_conv_transpose": (i32, [i32, vp, i32, i32, i32, vp, vp, i32, i32, i32, f32, vp, i32, vp, i32, cp, sz]),
        "kx_test_lstm": (i32, [i32, vp, i32, i32, i32] + [vp] * 9 + [cp, sz]),
        "kx_test_source": (i32, [i32, vp, i32, i32, vp, f32, u64, u64, i32, vp, cp, sz]),
        "kx_test_attention": (i32, [i32, vp, vp, i32, i32, vp, cp, sz]),
        "kx_test_lstm_fault": (i32, [i32]),
        "kx_test_conv1d_epilogue": (i32, [i32, vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, vp, i32, f32, f32, vp, vp,
                                          i32, cp, sz]),
        "kx_test_lstm_parts": (i32, [i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _test_lib = lib
    return lib


TEST_ABI_SYMBOLS = ["kx_test_conv1d", "kx_test_lstm", "kx_test_source", "kx_test_attention", "kx_test_conv1d_epilogue", "kx_test_conv1d_full",
                    "kx_test_conv_transpose", "kx_test_lstm_fault", "kx_test_lstm_parts"]

ABI_SYMBOLS = [
    "kx_version", "kx_init", "kx_create", "kx_import_onnx", "kx_create_from_device_blob", "kx_create_replicas", "kx_replicas_times", "kx_create_partition", "kx_destroy",
    "kx_last_error", "kx_last_error_copy", "kx_infer",
    "kx_free_audio", "kx_infer_device", "kx_sync", "kx_set_pinned_durations", "kx_warmup", "kx_arena_bytes", "kx_call_times", "kx_model_status", "kx_model_info", "kx_dispatcher_health", "kx_set_utterance_base", "kx_set_lanes",
    "kx_set_conv_mode", "kx_get_conv_mode", "kx_set_stft_variant", "kx_get_stft_variant",
    "kx_profile_enable", "kx_profile_read", "kx_profile_detail", "kx_profile_aux", "kx_diag_enable", "kx_diag_count", "kx_diag_get", "kx_set_act_prescale", "kx_set_voice_table", "kx_infer_voices",
    "kx_infer_packed", "kx_free_packed", "kx_dispatcher_create", "kx_dispatcher_create_warm", "kx_dispatcher_submit", "kx_dispatcher_submit_ex", "kx_dispatcher_model_batches",
    "kx_dispatcher_stats", "kx_dispatcher_failures", "kx_dispatcher_destroy", "kx_debug_tap",
]


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a) -> Optional[np.ndarray]:
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class HipKoko:
    """MI355X-native stand-in for `OrtKoko` (one instance per GPU, calls serialised)."""

    def __init__(self, model_path: str, device: int = 0, _handle=None):
        self._lib = load_library()
        self.device = device
        if _handle is not None:
            self._h = _handle
            return
        err = C.create_string_buffer(512)
        h = self._lib.kx_create(os.fsencode(model_path), device, err, len(err))
        if not h:
            # reference: `.expect("Failed to create Kokoro TTS model")`, koko.rs:572
            raise RuntimeError(f"Failed to create Kokoro TTS model: {err.value.decode()}")
        self._h = h

    # -- reference-shaped API -------------------------------------------------------------
    @classmethod
    def new(cls, model_path: str, device: int = 0) -> "HipKoko":
        return cls(model_path, device)

    @classmethod
    def from_device_blob(cls, data_ptr: int, n_bytes: int, device: int = 0) -> "HipKoko":
        """Build from a weight blob already in this GPU's memory (after the RCCL broadcast)."""
        lib = load_library()
        err = C.create_string_buffer(512)
        h = lib.kx_create_from_device_blob(C.c_void_p(data_ptr), n_bytes, device, err, len(err))
        if not h:
            raise RuntimeError(f"Failed to create Kokoro TTS model: {err.value.decode()}")
        return cls("", device, _handle=h)

    @classmethod
    def replicas(cls, model_path: str, devices: Sequence[int]) -> List["HipKoko"]:
        """One model per entry of `devices` from ONE read of the weight file: upload to devices[0], device-to-device
        fan-out to the others inside the library (kx_create_replicas).  The list is what `Dispatcher` takes; the
        reference's server is one process holding every GPU (kokorox-openai/src/lib.rs:370-439)."""
        lib = load_library()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        if dev.ndim != 1 or dev.shape[0] < 1:
            raise ValueError("replicas: at least one device id")
        out = (C.c_void_p * dev.shape[0])()
        err = C.create_string_buffer(512)
        rc = lib.kx_create_replicas(os.fsencode(model_path), _ptr(dev), dev.shape[0], out, err, len(err))
        if rc != 0:
            raise RuntimeError(f"Failed to create Kokoro TTS model: {err.value.decode()}")
        return [cls("", int(d), _handle=out[i]) for i, d in enumerate(dev)]

    @staticmethod
    def replicas_times():
        """ms from the last `replicas` call's entry: [file read (+ conversion), blob resident on every device, models built]."""
        out = (C.c_double * 3)()
        load_library().kx_replicas_times(out)
        return list(out)

    @classmethod
    def partition(cls, model_path: str, device: int, part: int, n_parts: int) -> "HipKoko":
        """A model confined to 1 / n_parts of the device's CUs (kx_create_partition): partitions run side by side on one GPU."""
        lib = load_library()
        err = C.create_string_buffer(512)
        h = lib.kx_create_partition(os.fsencode(model_path), device, part, n_parts, err, len(err))
        if not h:
            raise RuntimeError(f"Failed to create Kokoro TTS model: {err.value.decode()}")
        return cls("", device, _handle=h)

    def last_error(self) -> str:
        """Thread-safe copy of the model's last failure text (kx_last_error_copy)."""
        buf = C.create_string_buffer(512)
        self._lib.kx_last_error_copy(self._h, buf, len(buf))
        return buf.value.decode()

    def infer(self, tokens: Sequence[Sequence[int]], styles: Sequence[Sequence[float]], speed: float,
              seed: int = 0, flags: int = 0):
        """tokens [[...]] (already 0-wrapped, koko.rs:1169-1175), styles [[256 floats]], speed.

        Batch of one returns the waveform as a 1-D float32 array (what koko.rs:1179 flattens);
        a batch of B returns a list of B arrays."""
        if len(tokens) == 0 or len(tokens[0]) == 0:
            raise ValueError("infer: empty token list")
        if len(styles) != len(tokens):
            raise ValueError("infer: one style row per utterance is required")
        outs = self.infer_batch(tokens, styles, [float(speed)], seed=seed, flags=flags)
        return outs[0] if len(outs) == 1 else outs

    # -- batched / instrumented forms -------------------------------------------------------
    def infer_batch(self, tokens, styles, speeds, seed: int = 0, flags: int = 0) -> List[np.ndarray]:
        B = len(tokens)
        lens = np.array([len(t) for t in tokens], dtype=np.int32)
        stride = int(lens.max()) if B else 0
        ids = np.zeros((B, max(stride, 1)), dtype=np.int64)
        for b, t in enumerate(tokens):
            ids[b, : len(t)] = np.asarray(t, dtype=np.int64)
        st = _f32(np.asarray(styles, dtype=np.float32).reshape(B, -1))
        if B and st.shape[1] != STYLE_DIM:
            raise ValueError(f"infer: style rows must have {STYLE_DIM} floats")
        sp = _f32(np.asarray(speeds, dtype=np.float32).reshape(-1))
        out = C.POINTER(C.c_float)()
        out_lens = np.zeros(max(B, 1), dtype=np.int64)
        rc = self._lib.kx_infer(self._h, _ptr(ids), ids.shape[1], _ptr(lens), B, _ptr(st), _ptr(sp), sp.shape[0],
                                seed, flags, C.byref(out), _ptr(out_lens))
        self._check(rc)
        try:
            total = int(out_lens[:B].sum())
            flat = np.ctypeslib.as_array(out, shape=(max(total, 1),))[:total].copy()
        finally:
            self._lib.kx_free_audio(out)
        res, o = [], 0
        for b in range(B):
            res.append(flat[o: o + int(out_lens[b])])
            o += int(out_lens[b])
        return res

    # -- device voice table, packed output -----------------------------
    def set_voice_table(self, table: np.ndarray):
        """table [n_voices, 511, 1, 256] (or [n, 511, 256]) as built by load_voices / hf_cache.rs:284-309."""
        t = _f32(np.asarray(table, dtype=np.float32).reshape(-1, 511, 256))
        self._check(self._lib.kx_set_voice_table(self._h, _ptr(t), t.shape[0]))

    def _unpack(self, out, nbytes, nsamp, B, fmt):
        total = int(nbytes[:B].sum())
        raw = C.string_at(out, total) if total else b""
        self._lib.kx_free_packed(out)
        res, o = [], 0
        dt = np.int16 if fmt == PACK_PCM16_MONO else np.float32
        for b in range(B):
            a = np.frombuffer(raw, dtype=dt, count=int(nbytes[b]) // np.dtype(dt).itemsize, offset=o).copy()
            res.append(a.reshape(-1, 2) if fmt == PACK_F32_STEREO else a)
            o += int(nbytes[b])
        return res

    def _ids_lens(self, tokens):
        B = len(tokens)
        lens = np.array([len(t) for t in tokens], dtype=np.int32)
        ids = np.zeros((B, max(int(lens.max()), 1)), dtype=np.int64)
        for b, t in enumerate(tokens):
            ids[b, : len(t)] = np.asarray(t, dtype=np.int64)
        return ids, lens

    def infer_voices(self, tokens, voice_ids, weights, speeds=(1.0,), seed: int = 0, flags: int = 0,
                     fmt: int = 0) -> List[np.ndarray]:
        """Style rows are looked up / mixed on the GPU: voice_ids [B, max_mix] (-1 = unused), weights [B, max_mix]
        (the number after the dot in "af_sky.4+af_nicole.5"); max_mix == 1 is the single-voice copy."""
        ids, lens = self._ids_lens(tokens)
        B = len(tokens)
        v = np.ascontiguousarray(np.asarray(voice_ids, dtype=np.int32).reshape(B, -1))
        w = _f32(np.asarray(weights, dtype=np.float32).reshape(B, -1))
        sp = _f32(np.asarray(speeds, dtype=np.float32).reshape(-1))
        out = C.c_void_p()
        nbytes, nsamp = np.zeros(B, np.int64), np.zeros(B, np.int64)
        self._check(self._lib.kx_infer_voices(self._h, _ptr(ids), ids.shape[1], _ptr(lens), B, _ptr(v), _ptr(w),
                                              v.shape[1], _ptr(sp), sp.shape[0], seed, flags, fmt, C.byref(out),
                                              _ptr(nbytes), _ptr(nsamp)))
        return self._unpack(out, nbytes, nsamp, B, fmt)

    def infer_packed(self, tokens, styles, speeds=(1.0,), seed: int = 0, flags: int = 0, fmt: int = 0):
        ids, lens = self._ids_lens(tokens)
        B = len(tokens)
        st = _f32(np.asarray(styles, dtype=np.float32).reshape(B, STYLE_DIM))
        sp = _f32(np.asarray(speeds, dtype=np.float32).reshape(-1))
        out = C.c_void_p()
        nbytes, nsamp = np.zeros(B, np.int64), np.zeros(B, np.int64)
        self._check(self._lib.kx_infer_packed(self._h, _ptr(ids), ids.shape[1], _ptr(lens), B, _ptr(st), _ptr(sp),
                                              sp.shape[0], seed, flags, fmt, C.byref(out), _ptr(nbytes), _ptr(nsamp)))
        return self._unpack(out, nbytes, nsamp, B, fmt)

    def infer_device(self, d_ids: int, t_stride: int, lens_host: np.ndarray, d_styles: int, speeds_host: np.ndarray,
                     d_audio: int, audio_ld: int, d_frames: int, seed: int = 0, flags: int = 0) -> int:
        """Raw device-pointer form (bench.py); returns the audio row length that is needed."""
        lens_host = np.ascontiguousarray(lens_host, dtype=np.int32)
        speeds_host = _f32(speeds_host)
        need = C.c_int64(0)
        rc = self._lib.kx_infer_device(self._h, C.c_void_p(d_ids), t_stride, _ptr(lens_host), lens_host.shape[0],
                                       C.c_void_p(d_styles), _ptr(speeds_host), speeds_host.shape[0], seed, flags,
                                       C.c_void_p(d_audio), audio_ld, C.c_void_p(d_frames), C.byref(need))
        if rc != 0 and need.value > audio_ld:
            return int(need.value)
        self._check(rc)
        return int(need.value)

    def sync(self):
        self._check(self._lib.kx_sync(self._h))

    def set_pinned_durations(self, pattern: Optional[Sequence[int]]):
        if not pattern:
            self._check(self._lib.kx_set_pinned_durations(self._h, None, 0))
            return
        p = np.ascontiguousarray(pattern, dtype=np.int32)
        self._check(self._lib.kx_set_pinned_durations(self._h, _ptr(p), p.shape[0]))

    def warmup(self, B: int, n_tokens: int, frames_per_token: int = 8):
        """Size the arenas / result buffer for B utterances x n_tokens at frames_per_token frames per token (kx_warmup)."""
        self._check(self._lib.kx_warmup(self._h, B, n_tokens, frames_per_token))

    def arena_bytes(self):
        out = (C.c_int64 * 3)()
        self._check(self._lib.kx_arena_bytes(self._h, out))
        return list(out)

    def info(self):
        """kx_model_info as a dict: source variant, whether the arithmetic is the reference's for that file, conv mode, vocabulary,
        voices, CU partition."""
        out = (C.c_int64 * 8)()
        self._check(self._lib.kx_model_info(self._h, out))
        kinds = {0: "kxw container", 1: "onnx fp32", 2: "onnx fp16/bf16 (widened)", 3: "onnx 8-bit quantised (weights de-quantised)",
                 4: "onnx 4-bit quantised (weights de-quantised)", -1: "cached conversion", -2: "device blob"}
        return {"variant": int(out[0]), "variant_name": kinds.get(int(out[0]), "?"), "reference_arithmetic": bool(out[1]), "conv_mode": int(out[2]),
                "n_vocab": int(out[3]), "n_voices": int(out[4]), "cu_partition": int(out[5]), "cu_partitions": int(out[6]), "cus": int(out[7])}

    def status(self):
        """[recurrence in use (0 resident / 1 streaming), hand-off time-outs, clean forwards until the resident forms return, re-run calls]."""
        out = (C.c_int64 * 4)()
        self._check(self._lib.kx_model_status(self._h, out))
        return list(out)

    def call_times(self):
        """Host milestones of the last call in ms from its entry: [front queued, front done (the host wait), back planned, back queued]."""
        out = (C.c_double * 4)()
        self._check(self._lib.kx_call_times(self._h, out))
        return list(out)

    def set_conv_mode(self, mode: int):
        """0 = f32 MFMA, 1 = f16x3 split MFMA, 6 = f16f8 (default: f16x3 with the cross terms of the 7 / 11-tap convs on 8-bit
        MFMAs), 4 / 5 = f16 / bf16 single product (opt-in reduced precision)."""
        self._check(self._lib.kx_set_conv_mode(self._h, mode))

    def get_conv_mode(self) -> int:
        return int(self._lib.kx_get_conv_mode(self._h))

    def set_stft_variant(self, variant: int):
        """0 = the ONNX export's conv-based STFT pair (default), 1 = torch.stft / torch.istft semantics."""
        self._check(self._lib.kx_set_stft_variant(self._h, variant))

    def get_stft_variant(self) -> int:
        return int(self._lib.kx_get_stft_variant(self._h))

    def set_lanes(self, n: int):
        """1 = one stream; 4 = the independent chains of the back half side by side; 0 (default) = 4 up to batch 32,
        else 1.  Bit-identical results for every value."""
        self._check(self._lib.kx_set_lanes(self._h, n))

    def set_utterance_base(self, base: int):
        self._check(self._lib.kx_set_utterance_base(self._h, base))

    def profile_enable(self, on: bool):
        self._check(self._lib.kx_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        n, ms, fl = C.c_int64(0), C.c_double(0), C.c_double(0)
        self._check(self._lib.kx_profile_read(self._h, C.byref(n), C.byref(ms), C.byref(fl)))
        return int(n.value), float(ms.value), float(fl.value)

    def profile_detail(self) -> np.ndarray:
        """[n, 10] rows of the last profile_read: rows, Cin, taps, dil, stride, store, cols, flops, ms, algorithmic bytes."""
        n = C.c_int64(0)
        self._check(self._lib.kx_profile_detail(self._h, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 10), dtype=np.float64)
        self._check(self._lib.kx_profile_detail(self._h, _ptr(out), n.value, C.byref(n)))
        return out[: n.value]

    def profile_aux(self):
        """(launches, bytes) of the unfused InstanceNorm statistics passes since the last call."""
        n, by = C.c_int64(0), C.c_double(0)
        self._check(self._lib.kx_profile_aux(self._h, C.byref(n), C.byref(by)))
        return int(n.value), float(by.value)

    def diag_enable(self, on: bool):
        """Measure every conv input (after its AdaIN affine) during the following calls (real-weights report)."""
        self._check(self._lib.kx_diag_enable(self._h, 1 if on else 0))

    def diag_records(self):
        """[(layer, rows, Cin, taps, pre-scale exponent, absmax, rms, elements)] since diag_enable(True)."""
        n = C.c_int64(0)
        self._check(self._lib.kx_diag_count(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            name = C.create_string_buffer(128)
            v = np.zeros(7, dtype=np.float64)
            self._check(self._lib.kx_diag_get(self._h, i, name, len(name), _ptr(v)))
            out.append((name.value.decode(), int(v[0]), int(v[1]), int(v[2]), int(v[3]), float(v[4]), float(v[5]), float(v[6])))
        return out

    def set_act_prescale(self, layer: str, log2_scale: int):
        """f16x3: multiply this layer's transformed input by 2**log2_scale before the hi/lo split (undone exactly)."""
        self._check(self._lib.kx_set_act_prescale(self._h, layer.encode(), log2_scale))

    def tap(self, name: str, b: int = 0) -> np.ndarray:
        c, l = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.kx_debug_tap(self._h, name.encode(), b, None, 0, C.byref(c), C.byref(l)))
        out = np.empty((c.value, l.value), dtype=np.float32)
        self._check(self._lib.kx_debug_tap(self._h, name.encode(), b, _ptr(out), out.size, C.byref(c), C.byref(l)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise KokoroxHipError(rc, self.last_error())


class Dispatcher:
    """Batching front of one or more HipKoko models (one per GPU): the replacement for the reference's
    one-request-at-a-time `Mutex<Session>` (ort_koko.rs:78).  `submit` blocks and is thread-safe."""

    def __init__(self, models: Sequence[HipKoko], max_batch: int = 64, max_wait_us: int = 2000, warm: Optional[Sequence[int]] = None):
        """warm = (tokens, frames per token): one discarded forward of max_batch such utterances on every model first
        (kx_dispatcher_create_warm), so that no request ever pays for an arena growing."""
        self._lib = load_library()
        self._models = list(models)  # keep them alive
        arr = (C.c_void_p * len(self._models))(*[m._h for m in self._models])
        err = C.create_string_buffer(256)
        if warm:
            self._d = self._lib.kx_dispatcher_create_warm(arr, len(self._models), max_batch, max_wait_us, int(warm[0]), int(warm[1]), err, len(err))
        else:
            self._d = self._lib.kx_dispatcher_create(arr, len(self._models), max_batch, max_wait_us, err, len(err))
        if not self._d:
            raise RuntimeError(f"dispatcher: {err.value.decode()}")

    def submit(self, ids: Sequence[int], style: Sequence[float], speed: float = 1.0, seed: int = 0) -> np.ndarray:
        a = np.ascontiguousarray(ids, dtype=np.int64)
        st = _f32(np.asarray(style, dtype=np.float32).reshape(-1))
        if st.shape[0] != STYLE_DIM:
            raise ValueError(f"style row must have {STYLE_DIM} floats")
        out = C.POINTER(C.c_float)()
        n = C.c_int64(0)
        err = C.create_string_buffer(256)
        rc = self._lib.kx_dispatcher_submit(self._d, _ptr(a), a.shape[0], _ptr(st), float(speed), seed, C.byref(out),
                                            C.byref(n), err, len(err))
        if rc != 0:
            raise KokoroxHipError(rc, err.value.decode())
        try:
            return np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[: n.value].copy()
        finally:
            self._lib.kx_free_audio(out)

    def submit_ex(self, ids: Sequence[int], style: Optional[Sequence[float]] = None,
                  voices: Optional[Sequence] = None, speed: float = 1.0, seed: int = 0, fmt: int = 0) -> np.ndarray:
        """The request as the reference's servers make it: `style` = the 256-float row, OR `voices` = one voice id
        (int: single voice, row copy) or [(voice id, weight), ...] (a mix "a.4+b.5", koko.rs:1255-1306) into the
        device voice table of every model; `fmt` = 0 f32 mono, 1 f32 stereo, 2 PCM16.  Returns the packed samples."""
        a = np.ascontiguousarray(ids, dtype=np.int64)
        st = vid = w = None
        n_mix = 0
        if style is not None:
            st = _f32(np.asarray(style, dtype=np.float32).reshape(-1))
            if st.shape[0] != STYLE_DIM:
                raise ValueError(f"style row must have {STYLE_DIM} floats")
        elif isinstance(voices, (int, np.integer)):
            vid, n_mix = np.array([voices], dtype=np.int32), 1
        else:
            vid = np.array([v for v, _ in voices], dtype=np.int32)
            w = np.array([x for _, x in voices], dtype=np.float32)
            n_mix = len(vid)
        out, nb, ns = C.c_void_p(), C.c_int64(0), C.c_int64(0)
        err = C.create_string_buffer(256)
        rc = self._lib.kx_dispatcher_submit_ex(self._d, _ptr(a), a.shape[0], _ptr(st), _ptr(vid), _ptr(w), n_mix,
                                               float(speed), seed, fmt, C.byref(out), C.byref(nb), C.byref(ns), err, len(err))
        if rc != 0:
            raise KokoroxHipError(rc, err.value.decode())
        try:
            raw = C.string_at(out, nb.value)
        finally:
            self._lib.kx_free_audio(C.cast(out, C.POINTER(C.c_float)))
        if fmt == 2:
            return np.frombuffer(raw, dtype=np.int16).copy()
        arr = np.frombuffer(raw, dtype=np.float32).copy()
        return arr.reshape(-1, 2) if fmt == 1 else arr

    def stats(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._lib.kx_dispatcher_stats(self._d, C.byref(a), C.byref(b), C.byref(c))
        per = (C.c_int64 * len(self._models))()
        self._lib.kx_dispatcher_model_batches(self._d, per, len(self._models))
        rp, rt = C.c_int64(0), C.c_int64(0)
        self._lib.kx_dispatcher_failures(self._d, C.byref(rp), C.byref(rt))
        hl = (C.c_int32 * len(self._models))()
        mf, rq = C.c_int64(0), C.c_int64(0)
        self._lib.kx_dispatcher_health(self._d, hl, len(self._models), C.byref(mf), C.byref(rq))
        return {"requests": a.value, "batches": b.value, "max_batch": c.value, "batches_per_model": list(per),
                "replayed_requests": rp.value, "retried_batches": rt.value, "healthy": [int(v) for v in hl],
                "failed_models": mf.value, "requeued_requests": rq.value}

    def close(self):
        if getattr(self, "_d", None):
            self._lib.kx_dispatcher_destroy(self._d)
            self._d = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- stand-alone kernel hooks (tests) -----------------------------------------------------
def import_onnx(onnx_path: str, out_path: str) -> str:
    """`model.onnx` -> KXHIPW01 container through the LIBRARY's own reader (csrc/onnx_import.cpp; host only, no GPU):
    what kx_create does in memory when it is handed the .onnx path the reference passes (koko.rs:570-573)."""
    lib = load_library()
    err = C.create_string_buffer(2048)
    rc = lib.kx_import_onnx(os.fsencode(onnx_path), os.fsencode(out_path), err, len(err))
    if rc != KX_OK:
        raise KokoroxHipError(rc, err.value.decode(errors="replace"))
    return out_path


def _err_call(fn, *args):
    err = C.create_string_buffer(512)
    rc = fn(*args, err, len(err))
    if rc != 0:
        raise KokoroxHipError(rc, err.value.decode())


CONV_F32, CONV_F16X3, CONV_F16F8 = 0, 1, 6
STFT_ONNX, STFT_TORCH = 0, 1


def conv1d(x, w, bias=None, stride=1, pad=0, dil=1, transposed=False, act=0, slope=0.0, alpha=None, norm=None,
           device=0, mode=None, pre=False):
    """Run the MFMA conv kernel alone: x [B,Cin,L], w [Cout,Cin,k] ([Cin,Cout,k] if transposed).  pre: stage the input
    through a pre-split image (conv_f16x3_pre.hip; direct-A kernels only)."""
    lib = load_test_library()
    if mode is None:
        mode = CONV_F32 if os.environ.get("KOKOROX_CONV", "") == "f32" else CONV_F16X3
    x, w = _f32(x), _f32(w)
    B, Cin, L = x.shape
    k = w.shape[2]
    if transposed:
        Cout = w.shape[1]
        Lout = (L - 1) * stride - 2 * pad + k
    else:
        Cout = w.shape[0]
        Lout = (L + 2 * pad - dil * (k - 1) - 1) // stride + 1
    y = np.zeros((B, Cout, Lout), dtype=np.float32)
    bias, alpha, norm = _f32(bias), _f32(alpha), _f32(norm)
    _err_call(lib.kx_test_conv1d, device, _ptr(x), B, Cin, L, _ptr(w), _ptr(bias), Cout, k, stride, pad, dil,
              1 if transposed else 0, act, float(slope), _ptr(alpha), _ptr(norm), _ptr(y), Lout, mode | (0x100 if pre else 0))
    return y


def conv_transpose(x, w, bias=None, stride=6, act=1, slope=0.1, resid=None, up_off=0, mode=1, pre=False, device=0):
    """Polyphase transposed conv as Generator.ups runs it: x [B,Cin,L], w [Cin,Cout,2*stride] -> y [B,Cout,Lout + up_off]
    (+ resid; up_off = 1: output from column 1, column 0 = reflection of column 1)."""
    lib = load_test_library()
    x, w = _f32(x), _f32(w)
    B, Cin, L = x.shape
    Cout, k = w.shape[1], w.shape[2]
    assert k == 2 * stride
    pad = (k - stride) // 2
    Lout = (L - 1) * stride - 2 * pad + k
    y = np.zeros((B, Cout, Lout + (1 if up_off else 0)), dtype=np.float32)
    _err_call(lib.kx_test_conv_transpose, device, _ptr(x), B, Cin, L, _ptr(w), _ptr(_f32(bias)), Cout, stride, act, float(slope),
              _ptr(_f32(resid)), 1 if up_off else 0, _ptr(y), mode | (0x100 if pre else 0))
    return y


def conv1d_epilogue(x, w, bias=None, pad=0, dil=1, resid=None, y_init=None, out_mul=1.0, out_div=1.0, want_stats=False,
                    mode=1, device=0):
    """Stride-1 conv through the epilogue forms: y = (conv + bias + resid [+ y_init]) * out_mul / out_div, and the
    fused per-row (sum, sum of squares) when want_stats.  Returns y or (y, stats[B,Cout,2])."""
    lib = load_test_library()
    x, w = _f32(x), _f32(w)
    B, Cin, L = x.shape
    Cout, _, k = w.shape
    Lout = L + 2 * pad - dil * (k - 1)
    y = np.zeros((B, Cout, Lout), dtype=np.float32) if y_init is None else _f32(y_init).copy()
    st = np.zeros((B, Cout, 2), dtype=np.float32) if want_stats else None
    _err_call(lib.kx_test_conv1d_epilogue, device, _ptr(x), B, Cin, L, _ptr(w), _ptr(_f32(bias)), Cout, k, pad, dil,
              _ptr(_f32(resid)), 0 if y_init is None else 1, float(out_mul), float(out_div), _ptr(y), _ptr(st), mode)
    return (y, st) if want_stats else y


def conv1d_full(x, w, bias=None, pad=0, dil=1, act=0, slope=0.0, alpha=None, norm=None, resid=None, y_init=None,
                out_mul=1.0, out_div=1.0, want_stats=False, lens=None, pad_ld=False, flat=False, mode=1, device=0, pre=False):
    """Stride-1 conv with the fused input transform (AdaIN affine + leaky / snake) AND the epilogue forms, on a ragged
    batch (lens[b] valid input columns), with pad_ld on the model's padded rows, with flat through the flat list of live
    tiles the model gives the direct-A kernels.  Returns y or (y, stats[B,Cout,2])."""
    lib = load_test_library()
    x, w = _f32(x), _f32(w)
    B, Cin, L = x.shape
    Cout, _, k = w.shape
    Lout = L + 2 * pad - dil * (k - 1)
    y = np.zeros((B, Cout, Lout), dtype=np.float32) if y_init is None else _f32(y_init).copy()
    st = np.zeros((B, Cout, 2), dtype=np.float32) if want_stats else None
    ln = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    _err_call(lib.kx_test_conv1d_full, device, _ptr(x), B, Cin, L, _ptr(ln), (1 if pad_ld else 0) | (2 if flat else 0), _ptr(w), _ptr(_f32(bias)),
              Cout, k, pad, dil, act, float(slope), _ptr(_f32(alpha)), _ptr(_f32(norm)), _ptr(_f32(resid)),
              0 if y_init is None else 1, float(out_mul), float(out_div), _ptr(y), _ptr(st), mode | (0x100 if pre else 0))
    return (y, st) if want_stats else y


def lstm(x, params, device=0):
    """x [B,L,n_in]; params = (w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r) -> [B,L,512]."""
    lib = load_test_library()
    x = _f32(x)
    B, L, n_in = x.shape
    ps = [_f32(p) for p in params]
    y = np.zeros((B, L, 512), dtype=np.float32)
    _err_call(lib.kx_test_lstm, device, _ptr(x), B, L, n_in, *[_ptr(p) for p in ps], _ptr(y))
    return y


def attention(qkv, lens, device=0):
    """qkv [B,2304,T] (rows Q|K|V, 12 heads x 64 each), lens [B] -> ctx [B,768,T] (ALBERT self-attention)."""
    lib = load_test_library()
    qkv = _f32(qkv)
    B, C, T = qkv.shape
    assert C == 2304
    lens = np.ascontiguousarray(np.asarray(lens, dtype=np.int32))
    ctx = np.zeros((B, 768, T), dtype=np.float32)
    _err_call(lib.kx_test_attention, device, _ptr(qkv), _ptr(lens), B, T, _ptr(ctx))
    return ctx


def harmonic_source(f0, lin_w, lin_b, seed=0, utt_base=0, noise_off=False, device=0):
    """f0 [B,2F] -> har_source [B,600F]."""
    lib = load_test_library()
    f0 = _f32(f0)
    B, F2 = f0.shape
    lin_w = _f32(np.asarray(lin_w).reshape(-1))
    out = np.zeros((B, 300 * F2), dtype=np.float32)
    _err_call(lib.kx_test_source, device, _ptr(f0), B, F2, _ptr(lin_w), float(lin_b), seed, utt_base,
              1 if noise_off else 0, _ptr(out))
    return out
