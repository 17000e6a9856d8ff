"""Per-kernel attribution of the ragged (free-running) step against the pinned one, per frame of audio.
usage: ragged_attrib.py <prof_pinned_dir> <pinned.json> <prof_free_dir> <free.json> <out.txt>"""
import csv
import glob
import json
import os
import re
import sys


def load(prof_dir, bench_json):
    stats = glob.glob(os.path.join(prof_dir, "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    d = json.loads(open(bench_json).read().strip().splitlines()[-1])
    frames = d["config"]["frames"] * d["config"]["batch_per_gpu"]
    # (--durations free makes one more call first, to size the audio buffer: it stops after the front half - its token-axis
    # kernels are in the sums, which overstates the ragged step's token-axis share by a seventh; the frame axis is exact)
    n_steps = d["steps"] + d["warmup"]
    per = {}
    for r in rows:
        name = re.sub(r"\(.*", "", r["Name"])[:100]
        per[name] = per.get(name, 0.0) + float(r["TotalDurationNs"]) / n_steps
    return per, frames, d


def main(pp, pj, fp, fj, out):
    a, fa, da = load(pp, pj)
    b, fb, db = load(fp, fj)
    names = sorted(set(a) | set(b), key=lambda n: -(b.get(n, 0.0) / fb - a.get(n, 0.0) / fa))
    ta, tb = sum(a.values()) / fa, sum(b.values()) / fb
    lines = ["# ragged (free-running durations) vs pinned (equal lengths) step, GPU kernel time per frame of audio, by kernel",
             "# commands: tools/ragged_profile.sh (rocprofv3 --kernel-trace --stats, 5 timed + 2 warm-up steps each)",
             f"# pinned: {fa:.0f} frames/step, {da['ms_per_step']:.1f} ms/step wall; ragged: {fb:.0f} frames/step, {db['ms_per_step']:.1f} ms/step wall",
             f"# kernel time per frame: pinned {ta:.1f} ns, ragged {tb:.1f} ns ({tb / ta:.3f}x)", "",
             f"{'kernel':100s} {'pinned ns/frame':>16s} {'ragged ns/frame':>16s} {'diff':>9s} {'share of diff':>14s}"]
    for n in names:
        x, y = a.get(n, 0.0) / fa, b.get(n, 0.0) / fb
        if max(x, y) < 0.5:
            continue
        lines.append(f"{n:100s} {x:16.1f} {y:16.1f} {y - x:9.1f} {(y - x) / (tb - ta) * 100:13.1f}%")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main(*sys.argv[1:6])
