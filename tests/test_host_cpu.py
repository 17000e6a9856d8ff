"""CPU: host-side logic — weight container, caller-side input producers, C ABI surface, loud failure."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from kokorox_amd import voices as V
from kokorox_amd import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_param_count_and_spec():
    spec = W.tensor_spec()
    assert W.n_params() == 81_140_664  # model card 81.76 M minus the unused ALBERT pooler and weight-norm g's
    assert len(spec) == len(set(spec))
    assert spec["decoder.generator.ups.0.weight"][0] == (512, 256, 20)
    assert spec["bert.embeddings.word_embeddings.weight"][0] == (178, 128)


def test_blob_roundtrip(tmp_path):
    from collections import OrderedDict
    t = OrderedDict(a=np.arange(12, dtype=np.float32).reshape(3, 4), b=np.ones((5,), np.float32),
                    c=np.zeros((2, 3, 4), np.float32))
    p = str(tmp_path / "x.kxw")
    W.write_blob(p, t)
    r = W.read_blob(p)
    assert list(r) == ["a", "b", "c"]
    for k in t:
        np.testing.assert_array_equal(np.asarray(r[k]), t[k])
    with open(p, "r+b") as f:
        f.write(b"BADMAGIC")
    with pytest.raises(ValueError):
        W.read_blob(p)


def test_synthetic_weights_are_deterministic():
    a = W._init(np.random.default_rng(5), (4, 3), ("fan_in", 1.0, 3))
    b = W._init(np.random.default_rng(5), (4, 3), ("fan_in", 1.0, 3))
    np.testing.assert_array_equal(a, b)
    v = W.synthetic_voices(2)
    assert v.shape == (2, 511, 1, 256) and not v[:, 510].any()


def test_reference_token_rows(ref_inputs):
    """Known-answer rows of the reference's own tokenizer tests (tokenize.rs:120-129)."""
    sym = ref_inputs["symbols"]
    assert len(sym) == 178
    for text, ids in ref_inputs["token_rows"].items():
        assert V.tokenize(text, sym) == ids
    assert V.tokenize("Hi世!", sym) == V.tokenize("Hi!", sym)  # unknown chars are dropped
    assert V.pad_tokens([5, 6], initial_silence=2) == [[0, 30, 30, 5, 6, 0]]
    row = ref_inputs["ort_koko_sample_row"]
    assert row[0] == 0 and row[-1] == 0 and max(row) < 178


def test_mix_styles_matches_reference_rule(tmp_path):
    rng = np.random.default_rng(0)
    tab = {"af_sky": rng.standard_normal((510, 1, 256)).astype(np.float32),
           "af_nicole": rng.standard_normal((510, 1, 256)).astype(np.float32)}
    p = str(tmp_path / "voices.npz")
    np.savez(p, **tab)
    styles = V.load_voices(p)
    assert styles["af_sky"].shape == (511, 1, 256) and not styles["af_sky"][510].any()  # hf_cache.rs:302-309
    one = V.mix_styles(styles, "af_sky", 128)
    np.testing.assert_array_equal(np.float32(one[0]), tab["af_sky"][128, 0])
    mix = np.float32(V.mix_styles(styles, "af_sky.4+af_nicole.5", 17)[0])
    want = (tab["af_sky"][17, 0] * np.float32(np.float32(4) * np.float32(0.1))
            + tab["af_nicole"][17, 0] * np.float32(np.float32(5) * np.float32(0.1)))
    np.testing.assert_allclose(mix, want, rtol=1e-6)       # un-normalised: weights sum to 0.9
    # parts without '.' or with a non-numeric weight are skipped silently (koko.rs:1275-1285)
    np.testing.assert_array_equal(np.float32(V.mix_styles(styles, "af_sky.4+junk+af_nicole.x", 3)[0]),
                                  np.float32(V.mix_styles(styles, "af_sky.4+zzz", 3)[0]))
    with pytest.raises(KeyError):
        V.mix_styles(styles, "nobody", 3)
    with pytest.raises(KeyError):
        V.mix_styles(styles, "nobody.4+af_sky.5", 3)
    with pytest.raises(ValueError):
        V.mix_styles(styles, "a+b", 3)


def _header_symbols(name="kokorox_hip.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kx_[a-z0-9_]+)\s*\(", txt)))


def test_c_abi_exports_every_declared_symbol():
    from kokorox_amd import hip_koko as hk
    lib = hk.load_library()
    names = _header_symbols()
    assert "kx_infer" in names and "kx_create" in names and len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"libkokorox_hip.so does not export {n}"
    assert sorted(hk.ABI_SYMBOLS) == names
    assert b"gfx950" in lib.kx_version()


def test_test_hooks_live_in_their_own_library():
    """The stand-alone kernel hooks (include/kokorox_hip_test.h) are exported by libkokorox_hip_test.so and by it alone: the
    production library ships none of them."""
    from kokorox_amd import hip_koko as hk
    lib, tlib = hk.load_library(), hk.load_test_library()
    names = [n for n in _header_symbols("kokorox_hip_test.h") if n.startswith("kx_test_")]
    assert sorted(names) == sorted(hk.TEST_ABI_SYMBOLS) and len(names) >= 8
    for n in names:
        assert hasattr(tlib, n), f"libkokorox_hip_test.so does not export {n}"
        assert not hasattr(lib, n), f"libkokorox_hip.so exports the test hook {n}"
    assert not [n for n in _header_symbols() if n.startswith("kx_test_")]


def _split_args(arglist: str):
    """Top-level comma split of a parameter list (no nested parentheses in this ABI)."""
    a = [x.strip() for x in arglist.replace("\n", " ").split(",")]
    return [] if a in ([""], ["void"]) else a


def test_rust_crate_binds_the_header_name_for_name():
    """kokorox-hip/src/lib.rs (source only: no cargo in this image) declares every non-test entry point of
    include/kokorox_hip.h with the same number of parameters, and nothing the header lacks."""
    hdr = open(os.path.join(ROOT, "include", "kokorox_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    c_decl = {m.group(1): _split_args(m.group(2))
              for m in re.finditer(r"\b(kx_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", hdr)}
    rs = open(os.path.join(ROOT, "kokorox-hip", "src", "lib.rs"), encoding="utf-8").read()
    ext = re.search(r'extern "C" \{(.*?)\n\}', rs, flags=re.S).group(1)
    rs_decl = {m.group(1): _split_args(m.group(2)) for m in re.finditer(r"\bfn (kx_[a-z0-9_]+)\s*\(([^()]*)\)", ext)}
    assert len(rs_decl) >= 25
    for name, args in rs_decl.items():
        assert name in c_decl, f"lib.rs binds {name}, which include/kokorox_hip.h does not declare"
        assert len(args) == len(c_decl[name]), f"{name}: {len(args)} parameters in lib.rs, {len(c_decl[name])} in the header"
    product = {n for n in c_decl if not n.startswith("kx_test_") and n != "kx_debug_tap"}
    assert product <= set(rs_decl), f"lib.rs lacks {sorted(product - set(rs_decl))}"
    # pointer-vs-value agreement for every parameter (a `*` on one side means a `*` on the other)
    for name in product:
        for ca, ra in zip(c_decl[name], rs_decl[name]):
            assert ("*" in ca) == ("*" in ra), f"{name}: `{ca}` vs `{ra}`"
    for f in ("Cargo.toml", "build.rs"):
        assert os.path.exists(os.path.join(ROOT, "kokorox-hip", f))


def test_no_gpu_fails_loudly():
    """The product path has no CPU fallback: without a device, init/create report an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from kokorox_amd import hip_koko as hk
    lib = hk.load_library()
    err = C.create_string_buffer(256)
    assert lib.kx_init(0, err, len(err)) == 3 and b"no HIP device" in err.value
    with pytest.raises(RuntimeError, match="Failed to create Kokoro TTS model"):
        hk.HipKoko.new("/nonexistent.kxw")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "kokorox_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), encoding="utf-8").read()
                assert "import oracle" not in src and "from oracle" not in src, f


def _fake_upstream_checkpoint(tensors, rng):
    """Synthetic weights dressed as the upstream checkpoint: nested per module, 'module.' prefixes,
    weight-norm (g, v) pairs for the convs, plus tensors the forward never reads."""
    nested = {}
    for name, a in tensors.items():
        top, key = name.split(".", 1)
        sd = nested.setdefault(top, {})
        is_conv = key.endswith(".weight") and a.ndim == 3 and "embeddings" not in key and top in (
            "predictor", "text_encoder", "decoder") and "alpha" not in key and "noise_convs" not in key and \
            "m_source" not in key and "_proj" not in key
        if is_conv:
            v = (a.astype(np.float64) * rng.uniform(0.5, 2.0, size=(a.shape[0], 1, 1))).astype(np.float32)
            g = np.sqrt((a.astype(np.float64).reshape(a.shape[0], -1) ** 2).sum(1)).reshape(-1, 1, 1).astype(np.float32)
            sd["module." + key[:-7] + ".weight_g"] = g
            sd["module." + key[:-7] + ".weight_v"] = v
        else:
            sd["module." + key] = a
    nested["bert"]["module.embeddings.position_ids"] = np.arange(512, dtype=np.float32)[None]
    nested["bert"]["module.pooler.weight"] = np.zeros((768, 768), np.float32)
    nested["bert"]["module.pooler.bias"] = np.zeros(768, np.float32)
    nested["decoder"]["module.encode.norm1.norm.weight"] = np.ones(514, np.float32)
    nested["decoder"]["module.encode.norm1.norm.bias"] = np.zeros(514, np.float32)
    return nested


def test_importer_folds_weight_norm_and_matches_spec(tmp_path):
    """an upstream-style checkpoint becomes the blob kx_create reads."""
    from kokorox_amd import importer as I
    rng = np.random.default_rng(0)
    spec = W.tensor_spec()
    # small stand-in values with the real shapes (only a slice of the big tensors is random: speed)
    tensors = {}
    for name, (shape, _) in spec.items():
        a = np.zeros(shape, np.float32)
        a.reshape(-1)[: min(a.size, 4096)] = rng.standard_normal(min(a.size, 4096)).astype(np.float32)
        a.reshape(-1)[-1] = 1.0
        if a.ndim == 3:  # keep every conv row non-zero so its weight-norm is defined
            a[:, 0, 0] += 0.5
        tensors[name] = a
    ckpt = _fake_upstream_checkpoint(tensors, rng)
    out = I.to_blob_tensors(ckpt)
    assert list(out) == list(spec)
    for name in ("decoder.generator.resblocks.3.convs1.1.weight", "text_encoder.cnn.0.0.weight",
                 "decoder.generator.ups.0.weight", "predictor.lstm.weight_hh_l0_reverse", "bert_encoder.bias"):
        np.testing.assert_allclose(out[name], tensors[name], rtol=2e-6, atol=1e-7)
    p = str(tmp_path / "imported.kxw")
    W.write_blob(p, out)
    back = W.read_blob(p)
    np.testing.assert_array_equal(np.asarray(back["decoder.generator.conv_post.bias"]), out["decoder.generator.conv_post.bias"])
    # strictness: a missing tensor, a wrong shape and an unknown tensor are all errors
    bad = {k: dict(v) for k, v in ckpt.items()}
    del bad["bert_encoder"]["module.bias"]
    with pytest.raises(KeyError):
        I.to_blob_tensors(bad)
    bad = {k: dict(v) for k, v in ckpt.items()}
    bad["predictor"]["module.mystery.weight"] = np.zeros(3, np.float32)
    with pytest.raises(KeyError):
        I.to_blob_tensors(bad)
    bad = {k: dict(v) for k, v in ckpt.items()}
    bad["bert_encoder"]["module.weight"] = np.zeros((512, 767), np.float32)
    with pytest.raises(ValueError):
        I.to_blob_tensors(bad)


def test_importer_reads_safetensors(tmp_path):
    from safetensors.numpy import save_file
    from kokorox_amd import importer as I
    a = {"x.weight": np.arange(6, dtype=np.float32).reshape(2, 3), "y.bias": np.ones(4, np.float16)}
    p = str(tmp_path / "t.safetensors")
    save_file(a, p)
    r = I.read_safetensors(p)
    np.testing.assert_array_equal(r["x.weight"], a["x.weight"])
    np.testing.assert_array_equal(r["y.bias"], a["y.bias"].astype(np.float32))


def test_tts_chunks_builds_the_reference_calls():
    """per chunk id-30 prefix, style row picked by the chunk's own token count, 0 ... 0 padding,
    waveforms appended in order (koko.rs:1161-1180)."""
    from kokorox_amd import voices as V

    table = {"af": np.arange(511 * 256, dtype=np.float32).reshape(511, 1, 256)}
    calls = {}

    class FakeModel:
        def infer_batch(self, tokens, styles, speeds, seed=0, flags=0):
            calls["tokens"], calls["styles"], calls["speeds"], calls["seed"] = tokens, np.asarray(styles), speeds, seed
            return [np.full(len(t), float(i), dtype=np.float32) for i, t in enumerate(tokens)]

    out = V.tts_chunks(FakeModel(), table, "af", [[5, 6, 7], [9]], speed=1.25, initial_silence=2, seed=7)
    assert calls["tokens"] == [[0, 30, 30, 5, 6, 7, 0], [0, 30, 30, 9, 0]]
    assert np.array_equal(calls["styles"][0], table["af"][5, 0]) and np.array_equal(calls["styles"][1], table["af"][3, 0])
    assert calls["speeds"] == [1.25] and calls["seed"] == 7
    assert np.array_equal(out, np.concatenate([np.zeros(7, np.float32), np.ones(5, np.float32)]))
    assert V.tts_chunks(FakeModel(), table, "af", []).shape == (0,)
    with pytest.raises(ValueError):
        V.tts_chunks(FakeModel(), table, "af", [[]])


def test_xcd_aware_tile_orders_are_bijections():
    """The direct-A kernels map a workgroup's dispatch index to a tile so that each of the 8 XCDs gets a contiguous range of
    tiles (conv_f16x3_da.hip: the dense per-utterance slab form and, for ragged batches, the flat list over all live tiles).
    Restated here: both maps must hit every tile exactly once, whatever the grid size (a skipped or doubled tile would be a
    wrong result on the GPU)."""
    def flat(l, N):
        q, rem = N >> 3, N & 7
        cls = l & 7
        return cls * q + min(cls, rem) + (l >> 3)

    def dense(l, N, z):
        off = (N * z) & 7
        cls = (l + off) & 7
        start = 0
        for c in range(cls):
            first = (c - off) & 7
            start += (N - first + 7) >> 3 if first < N else 0
        return start + (l >> 3)

    for N in list(range(16, 600)) + [1170, 12672, 16896, 65537]:
        assert sorted(flat(l, N) for l in range(N)) == list(range(N)), N
        for z in (0, 1, 5):
            assert sorted(dense(l, N, z) for l in range(N)) == list(range(N)), (N, z)
