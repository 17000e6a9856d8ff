"""GPU: bench.py's contract — one JSON line with the required fields — at N=1 and, as a rehearsal of the
multi-rank path on a 1-GPU box, at N=2 over gloo with both ranks sharing the card (the real runs use
one GPU per rank over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _last_json(out: str):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--batch", "4", "--cpu-utts", "1",
                        "--free-run", "0"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["higher_is_better"] and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert "workload" in d["config"] and "model" not in d["config"]


def test_bench_two_ranks_rehearsal():
    env = dict(os.environ, KX_DIST_BACKEND="gloo", KX_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "1",
                        "--warmup", "1", "--batch", "4", "--cpu-utts", "0", "--free-run", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert abs(d["audio_s_per_step"] - 2 * 4 * 10.55) < 1e-6


def test_bench_blocks_and_single_process_replicas():
    """The secondary blocks of the N = 1 line (serving leg = BASELINE configs[4], reduced-precision mode = configs[2]'s
    "bf16") and the single-process form of the N-GPU run (`--replicas`, kx_create_replicas + one host thread per model),
    rehearsed with two replicas on the one GPU."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--batch", "4", "--cpu-utts", "0",
                        "--pcie", "0"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    fr = d["free_running"]
    assert fr["steps"] >= 5 and fr["min_frames"] <= fr["max_frames"] and fr["per_frame_vs_pinned"] > 0
    s = d["serve"]
    assert s["clients"] == 32 and s["requests"] >= 1000 and s["aggregate_rtf"] > 0 and s["latency_p99_ms"] >= s["latency_p50_ms"] > 0
    assert s["batches"] < s["requests"] and s["max_batch"] > 1 and s["wall_s"] > 1.0 and s["models"] == 1
    assert all(b > 0 for b in s["batches_per_model"])
    assert [o["offered_load"] for o in s["open_loop"]] == [0.5, 0.75, 0.9]
    for o in s["open_loop"]:
        assert o["requests"] >= 50 and o["latency_p99_ms"] >= o["latency_p50_ms"] > 0 and o["aggregate_rtf"] > 0
    lb = d["latency_b1"]
    assert lb["calls"] == 50 and 0 < lb["min_ms"] <= lb["median_ms"] <= lb["p99_ms"]
    rp = d["reduced_precision"]
    assert rp["value"] > 0 and rp["roofline"]["frac"] > 0 and "f16" in rp["mode"]
    assert "traffic_source" in d["roofline"]
    env = dict(os.environ, KX_REPLICA_IDS="0,0")
    r = subprocess.run([sys.executable, "bench.py", "--replicas", "2", "--steps", "1", "--warmup", "1", "--batch", "4", "--serve", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["weight_broadcast_s"] > 0
    # the single-process N-GPU line carries replica 0's roofline (HIP events on its own stream) and says where the CPU baseline is
    rl = d["roofline"]
    assert rl["bound"] == "mfma" and 0 < rl["frac"] < 1 and rl["achieved"] > 0 and "replica 0 of 2" in rl["measured_on"]
    assert d["cpu_baseline"] is None and "N = 1" in d["cpu_baseline_note"]
    assert d["replicas_times_ms"]["models_built"] >= d["replicas_times_ms"]["file_read"] > 0
    lb = _last_json(subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--batch", "4", "--cpu-utts", "0", "--pcie", "0",
                                    "--serve", "0", "--free-run", "0", "--reduced", "0", "--latency-b1", "10"], cwd=ROOT, capture_output=True,
                                   text=True, timeout=600).stdout)["latency_b1"]
    hm = lb["host_ms"]  # (kx_call_times: where a batch-1 call's wall time goes on the host)
    assert 0 < hm["front_queued"] <= hm["front_done"] <= hm["back_planned"] <= hm["back_queued"] <= lb["median_ms"]
    assert abs(d["audio_s_per_step"] - 2 * 4 * 10.55) < 1e-6 and "kx_create_replicas" in d["config"]["parallelism"]
