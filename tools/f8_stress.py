"""Determinism of the forward under memory contention: the ring loads of the direct-A kernels are waited for by hand-counted
s_waitcnt vmcnt(N); a wrong count would hand on a stale fragment only when memory is slow.  A background stream keeps HBM busy with
large copies while the same batch is run again and again: every run must give the bits of the first.
usage: python tools/f8_stress.py [runs] [batch]   (on the GPU box; ~20 s; batch 1 .. 2 exercises the small-grid tile forms)"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kokorox_amd import hip_koko as hk  # noqa: E402
from kokorox_amd import weights as W  # noqa: E402
from oracle import kokoro_ref as R  # noqa: E402  (inputs only)


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    m = hk.HipKoko.new(W.ensure_synthetic_blob())
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    ids = R.synthetic_inputs(B, 128, seed=0)
    voices = W.synthetic_voices(4)
    styles = [voices[b % 4, 128, 0] for b in range(B)]
    m.set_pinned_durations([3, 3, 3, 4])
    stop = False
    copied = [0]

    def hammer():
        s = torch.cuda.Stream()
        a = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
        b = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
        with torch.cuda.stream(s):
            while not stop:
                for _ in range(8):
                    b.copy_(a, non_blocking=True)
                    a.copy_(b, non_blocking=True)
                s.synchronize()
                copied[0] += 16

    ref = m.infer_batch([list(x) for x in ids], styles, [1.0], seed=2)
    th = threading.Thread(target=hammer)
    th.start()
    t0 = time.time()
    bad = 0
    try:
        for r in range(runs):
            out = m.infer_batch([list(x) for x in ids], styles, [1.0], seed=2)
            for a, b in zip(ref, out):
                if not np.array_equal(a, b):
                    bad += 1
    finally:
        stop = True
        th.join()
    print(f"conv mode {m.get_conv_mode()}: {runs} runs of batch {B} beside {copied[0] * 0.5:.0f} GiB of background copies in {time.time() - t0:.1f} s: "
          f"{bad} utterances differed from the first run; status {m.status()}")
    m.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
