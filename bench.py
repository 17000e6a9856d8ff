"""Headline benchmark: real-time factor of the Kokoro-82M forward at batch 64 per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (`OrtKoko::infer`'s replacement) over one batch of 64
synthetic 128-phoneme utterances per GPU (BASELINE.json configs[2]), inputs
already resident in HBM, durations pinned to 3,3,3,4 (F = 422 frames = 10.55 s of audio each).
Ranks hold different utterances (weak scaling); the only collective is the one-time
broadcast of the weight blob.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/F16 MFMA, dense (no sparsity)
_T0 = time.perf_counter()


def progress(msg: str):
    """Timestamped progress on stderr (a silent run is taken for a hang by the GPU runner)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def synthetic_ids(B: int, n_phonemes: int, seed: int) -> np.ndarray:
    """ids uniform on 1..177, wrapped with id 0 at both ends (koko.rs:1169-1173)."""
    rng = np.random.default_rng(seed)
    ids = rng.integers(1, 178, size=(B, n_phonemes), dtype=np.int64)
    z = np.zeros((B, 1), np.int64)
    return np.concatenate([z, ids, z], axis=1)


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.999)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, (q + per - 1) // per))
            break
        except Exception:
            continue
    return max(1, n)


def dtype_label(conv_mode: int) -> str:
    """kx_get_conv_mode: 0 f32 MFMA, 1 f16x3 split MFMA, 6 f16f8 (default), 4 / 5 reduced precision (KOKOROX_CONV=f16 / bf16)."""
    if conv_mode == 6:
        return ("f32 in / out, f32 accumulate; products on split operands: f16f8 (a_hi b_hi on f16 MFMAs, the cross terms a_lo b_hi + a_hi b_lo of "
                "the 7 / 11-tap convs on e4m3 scaled MFMAs: ~17 significant bits per product; every other conv f16x3: ~22) -- wider than the "
                "bf16 BASELINE configs[2] names, inside the 1e-4 parity band (same test as the f32 and f16x3 modes)")
    if conv_mode == 1:
        return "f32 (f16x3 split MFMA: 3 f16 MFMAs per product on hi/lo halves, f32 accumulate)"
    if conv_mode == 4:
        return ("f16 operands, f32 accumulate (KOKOROX_CONV=f16: one f16 MFMA per product in the decoder / generator convs; "
                "NARROWER than the reference's fp32 -- not the headline mode)")
    if conv_mode == 5:
        return ("bf16 operands, f32 accumulate (KOKOROX_CONV=bf16: one bf16 MFMA per product in the decoder / generator convs; "
                "NARROWER than the reference's fp32 -- not the headline mode)")
    return "f32"


def pmc_traffic(B, T, F, mode):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/*pmc_conv_traffic.json, written by tools/summarize_pmc.py for this same workload);
    None when no pass matches.  PMC passes cannot run inside this process."""
    import glob
    best, src = None, None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_conv_traffic*.json"))):
        try:
            d = json.load(open(p))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("batch"), w.get("tokens"), w.get("frames"), w.get("conv_mode", "f32")) == (B, T, F, mode):
            best, src = d, os.path.relpath(p, ROOT)
    return (None, None) if best is None else (best["traffic_bytes_per_launch"], src)


def roofline(f16x3, conv_flops, conv_ms, n_launch, steps, wall, B, T, F, conv_bytes=0.0, f8=False):
    """Dominant kernel = the BM=128 family of the conv1d implicit-GEMM kernel (HIP events in the library).

    achieved = ALGORITHMIC FLOPs (2*Cout*Cin*k*L per launch) / measured time.  In f16x3 mode every
    algorithmic multiply-add is issued as 3 f16 MFMAs, so the matrix pipe does 3x these FLOPs; the peak
    quoted is the plain dense f16 MFMA peak and `frac` is algorithmic/peak (pipe utilisation = 3*frac)."""
    if conv_ms <= 0:
        return None
    ach = conv_flops / (conv_ms * 1e-3) / 1e12
    peak = PEAK_F16_MFMA_TFLOPS if f16x3 else PEAK_F32_MFMA_TFLOPS
    kern = ("kx::conv1d_f16x3_da_kernel<ACT,K,NT,P1,W2,S16> + kx::conv1d_f16x3_dag_kernel + kx::conv1d_f16x3_kernel<128,..> (the 128-row "
            "f16x3 conv family, implicit GEMM, 3 f16 MFMAs per product: v_mfma_f32_32x32x16_f16, and v_mfma_f32_16x16x32_f16 "
            "in the S16 form that carries the 11-tap convs)" if f16x3
            else "kx::conv1d_mfma_kernel<128,128,2,2> (f32 32x32x2 MFMA implicit GEMM)")
    if f8:
        kern = ("kx::conv1d_f16x3_da_kernel<ACT,K,NT,P1,W2,S16,PRE,BF,F8> + kx::conv1d_f16x3_dag_kernel + kx::conv1d_f16x3_kernel<128,..> (the "
                "128-row conv family, implicit GEMM; f16f8 mode: the generator's 3 / 7 / 11-tap snake convs = 88 % of the family's FLOPs run v_mfma_f32_16x16x32_f16 "
                "for a_hi b_hi + v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3) for the two cross terms = 2 MFMA-equivalents per product, the rest 3 "
                "f16 MFMAs per product)")
    traffic, traffic_src = pmc_traffic(B, T, F, "f16f8" if f8 else ("f16x3" if f16x3 else "f32"))
    alg_bytes = conv_bytes / max(n_launch, 1)
    out = {"bound": "mfma", "kernel": kern, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
           "traffic": traffic,
           # (not measured by this run: PMC passes cannot run inside the process; the value is the committed pass below)
           "traffic_source": f"committed rocprofv3 --pmc pass {traffic_src} (tools/summarize_pmc.py)" if traffic_src else None,
           # algorithmic HBM bytes per launch (each conv reads its input, its residual / running sum
           # where the epilogue uses them and its weights once, and writes its output once; logged per launch by the
           # library) and how much more the PMC counters saw: > 1 = re-reads (window overlap, one staging per row tile)
           "algorithmic_bytes_per_launch": alg_bytes,
           "traffic_over_algorithmic": (traffic / alg_bytes) if (traffic and alg_bytes > 0) else None,
           "algorithmic_gb_per_s": alg_bytes / max(conv_ms / max(n_launch, 1) * 1e-3, 1e-12) / 1e9,
           "launches_per_step": n_launch / max(steps, 1), "avg_launch_ms": conv_ms / max(n_launch, 1),
           "gflop_per_launch": conv_flops / max(n_launch, 1) / 1e9, "kernel_share_of_wall": conv_ms * 1e-3 / wall}
    if f8:
        out["peak_note"] = "the dense f16 MFMA peak; the 8-bit instruction of the cross terms runs at twice that rate"
        out["mfma_issue_factor"] = "2 f16-MFMA-equivalents per product on the 7 / 11-tap convs, 3 elsewhere"
    elif f16x3:
        out["mfma_issue_factor"] = 3
        out["matrix_pipe_utilisation"] = 3 * ach / peak
    return out


def cpu_baseline(blob_path: str, n_utts: int, n_phonemes: int, pinned):
    """The CPU restatement (oracle/, torch fp32 on the host cores) on a bounded sample."""
    import torch
    from kokorox_amd import weights as W
    from oracle import kokoro_ref as R

    cores = usable_cores()
    torch.set_num_threads(cores)
    progress(f"cpu baseline: {cores} threads, loading oracle")
    o = R.KokoroOracle(blob_path)
    ids = synthetic_ids(n_utts + 1, n_phonemes, seed=0)
    voices = W.synthetic_voices(2)
    style = voices[0, n_phonemes, 0]
    o.forward(ids[0][:34], style, 1.0, seed=2, utt=0, pinned_dur=pinned[:34])  # warm-up (short)
    t = time.perf_counter()
    audio_s = 0.0
    for i in range(n_utts):
        a, _ = o.forward(ids[i + 1], style, 1.0, seed=2, utt=i, pinned_dur=pinned)
        audio_s += a.shape[0] / 24000.0
        progress(f"cpu baseline: utterance {i + 1}/{n_utts} done at {time.perf_counter() - t:.1f} s")
    wall = time.perf_counter() - t
    return {"value": audio_s / wall, "unit": "x realtime (audio-s/wall-s)", "cores": cores, "kind": "port",
            "sample": f"{n_utts} of the same 128-phoneme utterances, sequential batch-1 like the reference's "
                      f"Mutex<Session>, torch-CPU fp32 restatement (not ONNX Runtime), {wall:.1f} s wall",
            "utt_per_s": n_utts / wall}


def _serve_setup(models):
    from kokorox_amd import weights as W
    tab = W.synthetic_voices(4)  # af_sky, af_nicole, am_adam, bf_emma
    for m in models:
        m.set_pinned_durations(None)
        m.set_utterance_base(0)
        m.set_voice_table(tab)
    # three single voices, "af_sky.4+af_nicole.5" and "bf_emma.7+af_sky.3" (mix_styles, koko.rs:1255-1306)
    return [0, 1, [(0, 4.0), (1, 5.0)], 2, [(3, 7.0), (0, 3.0)]]


def _lat_summary(lat, audio, wall):
    lat = np.sort(np.asarray(lat))
    n = len(lat)
    return {"requests": int(n), "wall_s": wall, "audio_s": float(np.sum(audio)), "aggregate_rtf": float(np.sum(audio) / wall),
            "requests_per_s": n / wall, "latency_p50_ms": float(lat[n // 2] * 1e3),
            "latency_p99_ms": float(lat[min(n - 1, int(n * 0.99))] * 1e3), "latency_max_ms": float(lat[-1] * 1e3)}


def serve_leg(models, n_clients=32, per_client=48, max_batch=64, max_wait_us=3000, open_loop_s=5.0,
              open_loads=(0.5, 0.75, 0.9)):
    """BASELINE configs[4] (kokorox-openai/src/lib.rs:370-439): concurrent clients with mixed voices, incl. the mix
    "af_sky.4+af_nicole.5" named into the device voice table, over one dispatcher in front of `models`.

    Closed loop: `n_clients` clients, each sending its next request when the previous one is back (>= 1000 requests).
    Open loop: Poisson arrivals at three offered loads (fractions of the closed-loop request rate), `open_loop_s` seconds
    each; latency = completion - scheduled arrival, so queueing delay counts.  p50 / p99 / aggregate RTF per load."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from kokorox_amd import hip_koko as hk
    rules = _serve_setup(models)
    d = hk.Dispatcher(models, max_batch=max_batch, max_wait_us=max_wait_us)
    rng = np.random.default_rng(0)

    def make_request(r):
        k = int(r.integers(20, 129))
        ids = np.concatenate([[0], r.integers(1, 178, size=k), [0]]).astype(np.int64)
        return ids, rules[int(r.integers(0, len(rules)))], int(r.integers(0, 3)), int(r.integers(1, 2 ** 31))

    # untimed: as many concurrent full-length requests as there will be clients, twice, so that every model's arenas have met
    # the largest (batch x length) shape of the run (an arena that has to grow in the timed part is a stream sync + hipFree +
    # hipMalloc of gigabytes: one such regrowth was a 1.3 s latency outlier in a 25 s soak)
    warm = np.array([0] + [5] * 128 + [0], dtype=np.int64)
    with ThreadPoolExecutor(max_workers=max(n_clients, 2)) as ex:
        list(ex.map(lambda i: d.submit_ex(warm, voices=i % 3, seed=1 + i), range(2 * n_clients)))
    st0 = d.stats()
    errs = []

    # ---- closed loop ------------------------------------------------------------------------------------------------
    lat, audio = [], []
    lock = threading.Lock()

    def client(c):
        r = np.random.default_rng(100 + c)
        try:
            for _ in range(per_client):
                ids, voices, fmt, seed = make_request(r)
                t = time.perf_counter()
                w = d.submit_ex(ids, voices=voices, seed=seed, fmt=fmt)
                dt = time.perf_counter() - t
                with lock:
                    lat.append(dt)
                    audio.append(w.shape[0] / 24000.0)
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))

    t0 = time.perf_counter()
    th = [threading.Thread(target=client, args=(c,)) for c in range(n_clients)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    st1 = d.stats()
    if errs or not lat:
        d.close()
        return {"error": errs[:3]}
    closed = _lat_summary(lat, audio, wall)
    closed.update({"clients": n_clients, "batches": st1["batches"] - st0["batches"], "max_batch": st1["max_batch"],
                   "retried_batches": st1["retried_batches"], "replayed_requests": st1["replayed_requests"],
                   "batches_per_model": [b1 - b0 for b0, b1 in zip(st0["batches_per_model"], st1["batches_per_model"])]})
    progress(f"serve closed loop: {closed['requests']} requests in {wall:.2f} s, {closed['aggregate_rtf']:.0f}x, "
             f"p50/p99/max {closed['latency_p50_ms']:.0f}/{closed['latency_p99_ms']:.0f}/{closed['latency_max_ms']:.0f} ms, "
             f"{closed['retried_batches']} batches retried after a device failure")

    # ---- open loop: Poisson arrivals at fractions of the closed-loop rate ---------------------------------------------
    open_runs = []
    for load in open_loads:
        rate = load * closed["requests_per_s"]
        n_req = max(50, int(rate * open_loop_s))
        r = np.random.default_rng(int(load * 1000))
        arrivals = np.cumsum(r.exponential(1.0 / rate, size=n_req))
        reqs = [make_request(r) for _ in range(n_req)]
        lat_o, audio_o = [], []

        def fire(i, t_start):
            try:
                ids, voices, fmt, seed = reqs[i]
                w = d.submit_ex(ids, voices=voices, seed=seed, fmt=fmt)
                done = time.perf_counter()
                with lock:
                    lat_o.append(done - (t_start + arrivals[i]))
                    audio_o.append(w.shape[0] / 24000.0)
            except Exception as e:  # pragma: no cover
                errs.append(repr(e))

        b0 = d.stats()
        with ThreadPoolExecutor(max_workers=256) as ex:
            t_start = time.perf_counter()
            for i in range(n_req):
                dt = t_start + arrivals[i] - time.perf_counter()
                if dt > 0:
                    time.sleep(dt)
                ex.submit(fire, i, t_start)
        w_o = time.perf_counter() - t_start
        b1 = d.stats()
        if errs or not lat_o:
            break
        run = _lat_summary(lat_o, audio_o, w_o)
        run.update({"offered_load": load, "offered_requests_per_s": rate, "batches": b1["batches"] - b0["batches"],
                    "mean_batch": (b1["requests"] - b0["requests"]) / max(1, b1["batches"] - b0["batches"])})
        open_runs.append(run)
        progress(f"serve open loop {load:.2f}: {run['requests']} requests, p50/p99 {run['latency_p50_ms']:.0f}/"
                 f"{run['latency_p99_ms']:.0f} ms, {run['aggregate_rtf']:.0f}x")
    d.close()
    if errs:
        return {"error": errs[:3]}
    out = dict(closed)  # (the closed-loop figures stay at the top level: the keys earlier rounds reported)
    out.update({"models": len(models), "open_loop": open_runs,
                "workload": "20..128-phoneme requests, five voice rules (three single voices, af_sky.4+af_nicole.5, "
                            "bf_emma.7+af_sky.3) through the device voice table, f32 mono / f32 stereo / PCM16 mixed, free-running "
                            "durations, noise on; closed loop = each client waits for its answer; open loop = Poisson arrivals at "
                            "the stated fraction of the closed-loop request rate, latency from the scheduled arrival"})
    return out


def replicas_main(a):
    """Single-process form of the N-GPU run (`--replicas N`): kx_create_replicas reads the weight file once and fans the
    blob out over xGMI inside the library (the reference's server shape: one process holding every GPU), then one host
    thread per model steps its own utterances.  Same JSON keys as the torchrun form; `weight_broadcast_s` = the fan-out."""
    import threading
    import torch
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import weights as W
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ids_env = os.environ.get("KX_REPLICA_IDS")  # e.g. "0,0": rehearse two replicas on a one-GPU box
    dev_ids = [int(v) for v in ids_env.split(",")] if ids_env else list(range(a.replicas))
    N = len(dev_ids)
    if max(dev_ids) >= torch.cuda.device_count():
        raise SystemExit(f"--replicas {N} but only {torch.cuda.device_count()} GPU(s) visible")
    progress("preparing synthetic weight blob")
    blob_path = W.ensure_synthetic_blob()
    t0 = time.perf_counter()
    models = hk.HipKoko.replicas(blob_path, dev_ids)
    t_fan = time.perf_counter() - t0
    progress(f"{N} replicas ready ({t_fan:.2f} s incl. the one file read)")
    B, T = a.batch, a.phonemes + 2
    pinned = np.array([3, 3, 3, 4] * ((T + 3) // 4), dtype=np.int64)[:T]
    F = int(pinned.sum())
    audio_ld = 600 * F
    lens = np.full(B, T, dtype=np.int32)
    speeds = np.ones(1, dtype=np.float32)
    voices = W.synthetic_voices(4)
    state = []
    for r, (m, di) in enumerate(zip(models, dev_ids)):
        dev = torch.device("cuda", di)
        ids = torch.from_numpy(synthetic_ids(B, a.phonemes, seed=1000 + r)).to(dev)
        styles = torch.from_numpy(np.stack([voices[(r * B + b) % 4, a.phonemes, 0] for b in range(B)])).to(dev)
        audio = torch.empty((B, audio_ld), dtype=torch.float32, device=dev)
        frames = torch.zeros(B, dtype=torch.int32, device=dev)
        m.set_utterance_base(r * B)
        m.set_pinned_durations([3, 3, 3, 4])
        state.append((m, ids, styles, audio, frames))
    torch.cuda.synchronize()
    bar = threading.Barrier(N + 1)
    errs = []

    def worker(r):
        m, ids, styles, audio, frames = state[r]
        try:
            for _ in range(a.warmup):
                m.infer_device(ids.data_ptr(), T, lens, styles.data_ptr(), speeds, audio.data_ptr(), audio_ld, frames.data_ptr(), seed=2)
                m.sync()
            if r == 0:
                m.profile_enable(True)  # (replica 0's per-launch HIP events: the roofline block, as rank 0's in the torchrun form)
            bar.wait()  # start of the timed region
            for _ in range(a.steps):
                m.infer_device(ids.data_ptr(), T, lens, styles.data_ptr(), speeds, audio.data_ptr(), audio_ld, frames.data_ptr(), seed=2)
            m.sync()
            bar.wait()  # end: every replica has finished its K steps
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))
            bar.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(N)]
    for t in th:
        t.start()
    wall = 0.0
    try:
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    except threading.BrokenBarrierError:  # a worker failed and aborted the barrier: report ITS error below
        pass
    for t in th:
        t.join()
    if errs or wall <= 0.0:
        raise SystemExit(f"replica failed: {errs[0] if errs else 'barrier broken'}")
    for (m, ids, styles, audio, frames) in state:
        assert (frames.cpu().numpy() == F).all()
    finite = all(bool(torch.isfinite(s[3][:, : 600 * F]).all().item()) for s in state)
    audio_s_per_step = N * B * F * 600 / 24000.0
    # the dominant kernel's roofline from replica 0's own launches (HIP events on its stream), as rank 0 reports it under torchrun
    m0 = models[0]
    n_launch, conv_ms, conv_flops = m0.profile_read()
    det = m0.profile_detail()
    conv_bytes = float(det[:, 9].sum()) if len(det) else 0.0
    m0.profile_enable(False)
    rl = roofline(m0.get_conv_mode() in (1, 4, 5, 6), conv_flops, conv_ms, n_launch, a.steps, wall, B, T, F, conv_bytes, f8=m0.get_conv_mode() == 6)
    if rl is not None:
        rl["measured_on"] = f"replica 0 of {N} (device {dev_ids[0]}" + (f", CU partition 0 of {dev_ids.count(dev_ids[0])}" if dev_ids.count(dev_ids[0]) > 1 else "") + ")"
    out = {
        "metric": f"real-time factor (audio-s/wall-s), 24 kHz, batch={B}",
        "value": audio_s_per_step * a.steps / wall, "unit": "x realtime", "n_gpus": N, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype_label(models[0].get_conv_mode()),
        "data": "synthetic (seeded random-init Kokoro-82M weights, uniform phoneme ids, N(0,0.1) voice rows)",
        "config": {"workload": f"batch={B}/GPU synthetic {a.phonemes}-phoneme utterances (T={T}), durations pinned "
                               f"3,3,3,4 -> F={F} frames = {F * 600 / 24000.0:.2f} s each, noise on, inputs in HBM",
                   "batch_per_gpu": B, "tokens": T, "frames": F,
                   "parallelism": f"utterance-sharded x{N}, single process: kx_create_replicas (one file read, peer fan-out), "
                                  f"one host thread per model, device ids {dev_ids}"},
        "utterances_per_s": N * B * a.steps / wall, "audio_s_per_step": audio_s_per_step, "finite": finite,
        "weight_broadcast_s": t_fan, "replicas_times_ms": dict(zip(("file_read", "blob_resident_everywhere", "models_built"), hk.HipKoko.replicas_times())),
        "roofline": rl,
        "cpu_baseline": None, "cpu_baseline_note": "the CPU restatement is timed at N = 1 only (python bench.py); see BENCH_r*.json",
        "serve": serve_leg(models) if a.serve else None,
    }
    print(json.dumps(out), flush=True)
    for m in models:
        m.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--phonemes", type=int, default=128)
    ap.add_argument("--cpu-utts", type=int, default=12, help="utterances in the CPU baseline sample (0 = skip)")
    ap.add_argument("--detail", default="", help="write a per-shape table of the conv launches to this file")
    ap.add_argument("--free-run", type=int, default=1, help="also time one step with predicted durations")
    ap.add_argument("--pcie", type=int, default=1, help="also time one step through the host-buffer boundary (N=1)")
    ap.add_argument("--reduced", type=int, default=1,
                    help="also time the opt-in reduced-precision mode (one f16 MFMA per product) as a secondary block (N=1)")
    ap.add_argument("--serve", type=int, default=1, help="also run the 32-client serving leg (configs[4]) on rank 0 at N=1")
    ap.add_argument("--durations", default="pinned", choices=["pinned", "free"],
                    help="free = the timed steps run on the predicted (ragged) durations: a profiling mode (rocprof of the "
                         "ragged step, tools/ragged_profile.sh); the headline is the pinned workload")
    ap.add_argument("--serve-models", type=int, default=1,
                    help="models per GPU behind the serving leg's dispatcher: n > 1 = n CU-partitioned models (kx_create_replicas with the "
                         "device id given n times: each confined to 1 / n of the CUs, forwards side by side; DESIGN.md section 7)")
    ap.add_argument("--latency-b1", type=int, default=50, help="batch-1 calls timed for the latency_b1 block (configs[1]; 0 = skip)")
    ap.add_argument("--replicas", type=int, default=0,
                    help="single-process form: N models from kx_create_replicas, one host thread each (instead of torchrun)")
    a = ap.parse_args()
    if a.replicas > 0:
        return replicas_main(a)

    import faulthandler
    faulthandler.dump_traceback_later(300, repeat=True, file=sys.stderr)
    import torch
    import torch.distributed as dist
    from kokorox_amd import dist as kd
    from kokorox_amd import hip_koko as hk
    from kokorox_amd import weights as W

    rank, local_rank, world = kd.env_rank_world()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one process per GPU; KX_DIST_BACKEND=gloo + KX_SHARE_GPU=1 rehearse the N>1 path on a 1-GPU box
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % n_dev if os.environ.get("KX_SHARE_GPU") == "1" else local_rank
    if dev_index >= n_dev:
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("KX_DIST_BACKEND", "nccl")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- weights: rank 0 owns the file, everyone else gets it over RCCL/xGMI once ----------
    progress("preparing synthetic weight blob")
    blob_path = W.ensure_synthetic_blob() if rank == 0 else ""
    t0 = time.perf_counter()
    blob = kd.broadcast_blob(blob_path, dev, rank, world)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    progress(f"blob on device ({t_bcast:.2f} s); building model")
    model = hk.HipKoko.from_device_blob(blob.data_ptr(), blob.numel(), device=dev_index)
    del blob
    progress("model ready")

    # ---- this rank's utterances, resident in HBM -------------------------------------------
    B, T = a.batch, a.phonemes + 2
    ids = torch.from_numpy(synthetic_ids(B, a.phonemes, seed=1000 + rank)).to(dev)
    voices = W.synthetic_voices(4)
    styles = torch.from_numpy(np.stack([voices[(rank * B + b) % 4, a.phonemes, 0] for b in range(B)])).to(dev)
    lens = np.full(B, T, dtype=np.int32)
    speeds = np.ones(1, dtype=np.float32)
    pinned = np.array([3, 3, 3, 4] * ((T + 3) // 4), dtype=np.int64)[:T]
    F = int(pinned.sum())
    audio_ld = 600 * F
    audio = torch.empty((B, audio_ld), dtype=torch.float32, device=dev)
    frames = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    model.set_utterance_base(rank * B)
    model.set_pinned_durations([3, 3, 3, 4])
    free_main = a.durations == "free"
    if free_main:  # profiling mode: ragged batch in the timed region
        model.set_pinned_durations(None)
        need = model.infer_device(ids.data_ptr(), T, lens, styles.data_ptr(), speeds, audio.data_ptr(), audio_ld, frames.data_ptr(), seed=2)
        if need > audio_ld:
            audio_ld = int(need)
            audio = torch.empty((B, audio_ld), dtype=torch.float32, device=dev)
        a.free_run = a.pcie = a.reduced = a.serve = a.latency_b1 = 0

    def step():
        need = model.infer_device(ids.data_ptr(), T, lens, styles.data_ptr(), speeds, audio.data_ptr(), audio_ld,
                                  frames.data_ptr(), seed=2)
        assert need <= audio_ld, (need, audio_ld)

    def fence():
        model.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        step()
        model.sync()
        progress(f"warmup step {i + 1}/{a.warmup} done")
    fence()
    model.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    wall = time.perf_counter() - t0
    progress(f"{a.steps} timed steps: {wall:.3f} s")
    n_launch, conv_ms, conv_flops = model.profile_read()
    stats_launches, stats_bytes = model.profile_aux()
    conv_mode = model.get_conv_mode()
    f16x3 = conv_mode in (1, 4, 5, 6)  # all run the 16-bit matrix pipe (modes 4 / 5 = the opt-in reduced precision, labelled as such)
    det = model.profile_detail()
    conv_bytes = float(det[:, 9].sum()) if len(det) else 0.0
    if a.detail and rank == 0:
        per = len(det) // max(a.steps, 1)
        agg = {}
        for r in det[-per:]:
            key = tuple(int(v) for v in r[:6])
            t = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
            t[0] += 1; t[1] += r[7]; t[2] += r[8]; t[3] += r[9]
        with open(a.detail, "w") as f:
            f.write("rows Cin taps dil stride store launches GFLOP ms TFLOP/s alg_GB alg_GB/s\n")
            for key, (n, fl, ms, by) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
                f.write(" ".join(f"{v:5d}" for v in key) + f" {n:4d} {fl / 1e9:10.1f} {ms:9.3f} {fl / ms / 1e9:8.1f}"
                        f" {by / 1e9:8.2f} {by / ms / 1e6:8.0f}\n")
    model.profile_enable(False)
    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())

    fr = frames.cpu().numpy()
    if free_main:
        finite = all(bool(torch.isfinite(audio[b, : 600 * int(fr[b])]).all().item()) for b in range(B))
        audio_s_per_step = world * float(fr.sum()) * 600 / 24000.0  # (rank 0's frames stand for every rank's)
    else:
        assert (fr == F).all(), fr
        a_host = audio[:, : 600 * F].float()
        finite = bool(torch.isfinite(a_host).all().item())
        audio_s_per_step = world * B * F * 600 / 24000.0
    rtf = audio_s_per_step * a.steps / wall
    utt_s = world * B * a.steps / wall

    free = None
    if a.free_run:
        model.set_pinned_durations(None)
        cap = 600 * 50 * 4  # generous row: retried below if the prediction is longer
        buf = torch.empty((B, cap), dtype=torch.float32, device=dev)
        need = model.infer_device(ids.data_ptr(), T, lens, styles.data_ptr(), speeds, buf.data_ptr(), cap,
                                  frames.data_ptr(), seed=2)
        if need > cap:
            cap = need
            buf = torch.empty((B, cap), dtype=torch.float32, device=dev)
        # untimed pass so that the arenas have grown to this shape before the timed one
        model.infer_device(ids.data_ptr(), T, lens, styles.data_ptr(), speeds, buf.data_ptr(), cap,
                           frames.data_ptr(), seed=2)
        fence()
        n_free = max(5, a.steps // 2)
        t1 = time.perf_counter()
        for _ in range(n_free):
            model.infer_device(ids.data_ptr(), T, lens, styles.data_ptr(), speeds, buf.data_ptr(), cap,
                               frames.data_ptr(), seed=2)
        fence()
        w1 = (time.perf_counter() - t1) / n_free
        fr_free = frames.cpu().numpy()
        sf = int(fr_free.sum())
        progress(f"free-running steps: {w1:.3f} s each, sum frames {sf}")
        pinned_us_per_frame = wall / a.steps / (B * F) * 1e6
        free = {"steps": n_free, "sum_frames_rank0": sf, "min_frames": int(fr_free.min()), "max_frames": int(fr_free.max()),
                "audio_s_rank0": sf * 600 / 24000.0, "wall_s": w1, "rtf_rank0": sf * 600 / 24000.0 / w1,
                # ragged batch against the pinned (equal-length) one, per frame of audio
                "us_per_frame": w1 / sf * 1e6, "pinned_us_per_frame": pinned_us_per_frame,
                "per_frame_vs_pinned": (w1 / sf * 1e6) / pinned_us_per_frame}
        del buf

    # ---- the same batch through the host-buffer boundary (kx_infer: H2D of ids/styles, forward, D2H of the audio) ----
    pcie = None
    if a.pcie and rank == 0 and world == 1:
        model.set_pinned_durations([3, 3, 3, 4])
        model.set_utterance_base(rank * B)
        # timed at the C ABI itself (kx_infer with host pointers in, malloc'd host waveforms out, kx_free_audio): what
        # a Rust / C++ host pays; the numpy copies a Python caller adds on top are not the library's
        import ctypes as C
        ids_h = np.ascontiguousarray(ids.cpu().numpy())
        st_h = np.ascontiguousarray(styles.cpu().numpy())
        out_lens = np.zeros(B, dtype=np.int64)

        def host_call():
            out = C.POINTER(C.c_float)()
            rc = model._lib.kx_infer(model._h, ids_h.ctypes.data_as(C.c_void_p), T, lens.ctypes.data_as(C.c_void_p), B,
                                     st_h.ctypes.data_as(C.c_void_p), speeds.ctypes.data_as(C.c_void_p), 1, 2, 0,
                                     C.byref(out), out_lens.ctypes.data_as(C.c_void_p))
            assert rc == 0, model.last_error()
            first = float(out[0])
            model._lib.kx_free_audio(out)
            return first

        host_call()  # untimed: I/O arena and the pooled page-locked buffer at this shape
        t2 = time.perf_counter()
        host_call()
        w2 = time.perf_counter() - t2
        a_s = float(out_lens.sum()) / 24000.0
        pcie = {"rtf": a_s / w2, "ms_per_step": w2 * 1e3, "audio_s": a_s,
                "note": "kx_infer at the C ABI: pageable host ids/styles in, forward, the packed batch back by one async "
                        "D2H into pooled page-locked host memory, kx_free_audio"}
        progress(f"host-buffer step: {w2:.3f} s")

    # ---- secondary: the opt-in reduced-precision mode (BASELINE configs[2] says "bf16"; the reference's model_fp16 /
    # quantised variants, hf_cache.rs:135-144).  Never `value`: narrower than the reference's fp32 default. ----
    reduced = None
    f32_class = None
    if a.reduced and rank == 0 and world == 1 and conv_mode in (1, 6):
        model.set_pinned_durations([3, 3, 3, 4])
        model.set_utterance_base(rank * B)

        def reduced_run(mode, label):
            model.set_conv_mode(mode)
            try:
                step()
                model.sync()
                model.profile_enable(True)
                t3 = time.perf_counter()
                for _ in range(a.steps):
                    step()
                fence()
                w3 = time.perf_counter() - t3
                n3, ms3, fl3 = model.profile_read()
                model.profile_enable(False)
            finally:
                model.set_conv_mode(conv_mode)
            # (the family mixes launches at one MFMA per product -- the decoder / generator direct-A convs -- with launches
            # at three: `achieved` is algorithmic FLOPs over time, the issue factor is stated, no pipe utilisation is derived)
            ach3 = fl3 / (ms3 * 1e-3) / 1e12 if ms3 > 0 else 0.0
            progress(f"{'other-mode' if mode == 1 else 'reduced-precision'} steps ({label}): {w3:.3f} s")
            return {"value": audio_s_per_step * a.steps / w3, "unit": "x realtime", "ms_per_step": w3 / a.steps * 1e3,
                    "roofline": {"bound": "mfma", "achieved": ach3, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "frac": ach3 / PEAK_F16_MFMA_TFLOPS, "mfma_issue_factor": "1 on the direct-A convs, 3 elsewhere",
                                 "avg_launch_ms": ms3 / max(n3, 1), "launches_per_step": n3 / max(a.steps, 1)}}

        if conv_mode == 6:
            # the same steps with three f16 MFMAs per product everywhere (KOKOROX_CONV=f16x3, the default of rounds 2-4): what the
            # 8-bit cross terms buy, and the figure to compare with earlier rounds
            f32_class = reduced_run(1, "f16x3")
            f32_class["roofline"]["mfma_issue_factor"] = 3
            f32_class["mode"] = ("KOKOROX_CONV=f16x3: every product as three f16 MFMAs on hi / lo halves (~22 significant bits); the "
                                 "default mode's waveform differs from it by < 1e-5")
        reduced = reduced_run(4, "f16")
        reduced["mode"] = ("KOKOROX_CONV=f16: one v_mfma_f32_32x32x16_f16 per product in the decoder / generator convs of the "
                           "direct-A kernel (f16 operands, f32 accumulate); duration head, F0/N predictor, source, STFT f32-class")
        # the dtype BASELINE configs[2] names, on the same switch: one v_mfma_f32_32x32x16_bf16 per product, bf16 weight image
        bf = reduced_run(5, "bf16")
        bf["mode"] = "KOKOROX_CONV=bf16: the same with bf16 operands (v_mfma_f32_32x32x16_bf16): 8 significant bits instead of 11"
        reduced["bf16"] = bf
        reduced["note"] = ("secondary figures, never `value`: both leave the 1e-4 parity band; the waveform error of each against the "
                           "oracle is measured and bounded in tests/test_gpu_forward.py (test_reduced_precision_mode_error_is_bounded)")

    # ---- BASELINE configs[1]: one 128-phoneme utterance, fp32-class default mode, batch 1: latency of a call ----------
    lat_b1 = None
    if a.latency_b1 and rank == 0 and world == 1:
        model.set_pinned_durations([3, 3, 3, 4])
        model.set_utterance_base(rank * B)
        ids1, st1, lens1 = ids[:1].contiguous(), styles[:1].contiguous(), lens[:1].copy()
        a1 = torch.empty((1, audio_ld), dtype=torch.float32, device=dev)
        f1 = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()

        host_ms = []  # host-side milestones of every timed call (kx_call_times) + the moment kx_infer_device returned

        def one():
            t = time.perf_counter()
            model.infer_device(ids1.data_ptr(), T, lens1, st1.data_ptr(), speeds, a1.data_ptr(), audio_ld, f1.data_ptr(), seed=2)
            t_ret = time.perf_counter() - t
            model.sync()
            dt = time.perf_counter() - t
            host_ms.append(model.call_times() + [t_ret * 1e3])
            return dt

        for _ in range(5):
            one()
        host_ms.clear()
        # (the harness's own garbage collector off while the calls are timed: a collection between the library's return and
        # ctypes' -- 36 ms, one call in a few hundred, placed by tools/b1_outliers.py with kx_call_times -- is not the library's latency)
        import gc
        gc.collect()
        gc.disable()
        try:
            ts = np.sort(np.array([one() for _ in range(a.latency_b1)]))
        finally:
            gc.enable()
        hm = np.median(np.array(host_ms), axis=0)
        lat_b1 = {"calls": int(len(ts)), "median_ms": float(ts[len(ts) // 2] * 1e3), "p99_ms": float(ts[min(len(ts) - 1, int(len(ts) * 0.99))] * 1e3),
                  "min_ms": float(ts[0] * 1e3), "rtf_at_median": F * 600 / 24000.0 / float(ts[len(ts) // 2]),
                  # where a call's wall time goes on the host (medians, ms from the call's entry): the front half is queued, the GPU
                  # has finished it (the one host wait: predicted frame counts size the back half), the back half is planned, the
                  # back half is queued = kx_infer_device returns; the rest of the latency is kx_sync waiting for the GPU
                  "host_ms": {"front_queued": float(hm[0]), "front_done": float(hm[1]), "back_planned": float(hm[2]),
                              "back_queued": float(hm[3]), "infer_device_returned": float(hm[4])},
                  "workload": f"batch 1, one {a.phonemes}-phoneme utterance (T={T}), durations pinned 3,3,3,4 -> F={F} "
                              f"({F * 600 / 24000.0:.2f} s of audio), inputs in HBM, kx_infer_device + kx_sync per call"}
        progress(f"batch-1 latency: median {lat_b1['median_ms']:.2f} ms, p99 {lat_b1['p99_ms']:.2f} ms")

    if rank == 0:
        flops_per_utt = (0.1635 * T + 1.317 * F) * 1e9  # SURVEY.md §8d model
        out = {
            "metric": f"real-time factor (audio-s/wall-s), 24 kHz, batch={B}",
            "value": rtf,
            "unit": "x realtime",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": wall / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype_label(conv_mode),
            "data": "synthetic (seeded random-init Kokoro-82M weights, uniform phoneme ids, N(0,0.1) voice rows)",
            "config": {
                "workload": (f"batch={B}/GPU synthetic {a.phonemes}-phoneme utterances (T={T}), durations pinned "
                             f"3,3,3,4 -> F={F} frames = {F * 600 / 24000.0:.2f} s each, noise on, inputs in HBM") if not free_main else
                            (f"PROFILING MODE --durations free: batch={B}/GPU synthetic {a.phonemes}-phoneme utterances, PREDICTED "
                             f"durations (ragged: {int(fr.min())}..{int(fr.max())} frames, sum {int(fr.sum())}), noise on, inputs in HBM"),
                "batch_per_gpu": B, "tokens": T, "frames": F if not free_main else int(fr.sum()) / B, "parallelism": f"utterance-sharded x{world}, "
                "one-time weight broadcast",
            },
            "utterances_per_s": utt_s,
            "audio_s_per_step": audio_s_per_step,
            "finite": finite,
            "weight_broadcast_s": t_bcast,
            "model_tflops": flops_per_utt * world * B * a.steps / wall / 1e12,
            "roofline": roofline(f16x3, conv_flops, conv_ms, n_launch, a.steps, wall, B, T, F, conv_bytes, f8=conv_mode == 6),
            # unfused InstanceNorm statistics passes of the timed steps: each reads its tensor once (the known byte
            # count tools/summarize_pmc.py checks FETCH_SIZE against)
            "in_stats": {"launches_per_step": stats_launches / max(a.steps, 1), "bytes_per_step": stats_bytes / max(a.steps, 1)},
            "free_running": free,
            "latency_b1": lat_b1,
            "pcie_inclusive": pcie,
            "reduced_precision": reduced,
            "f16x3_mode": f32_class,
        }
        out["serve"] = None
        if world == 1 and a.serve:
            if a.serve_models > 1:
                # several models on the one GPU: CU-partitioned (each confined to 1 / n of the CUs, forwards side by side)
                serve_models = hk.HipKoko.replicas(blob_path, [dev_index] * a.serve_models)
            else:
                serve_models = [model]
            try:
                out["serve"] = serve_leg(serve_models)
            finally:
                if a.serve_models > 1:
                    for m2 in serve_models:
                        m2.close()
        if world == 1 and a.cpu_utts > 0:
            out["cpu_baseline"] = cpu_baseline(blob_path, a.cpu_utts, a.phonemes, pinned)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    model.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
