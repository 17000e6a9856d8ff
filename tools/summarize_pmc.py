"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) for the dominant kernel family.

Units/corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM exactly: the counters are in KiB; on gfx950
FETCH_SIZE reports HALF of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is taken as is.
No hand-fed constants: the doubling is CHECKED (not calibrated) on in_stats_kernel, which reads each tensor it
normalises exactly once - the bytes it must read come from the library's own launch accounting (bench.py prints them as
"in_stats" in its JSON line, which the FETCH_SIZE pass keeps as its stdout).

Per instantiation (round 4): with a `--detail` table of the same workload (bench.py --detail: algorithmic bytes per conv
shape) the traffic of every kernel of the family is set against the algorithmic bytes of the shapes that kernel runs
(grouped by tap count and by snake-resblock / other, which is what selects the instantiation).

usage: summarize_pmc.py <fetch_dir> <write_dir> <out.json> <bench stdout of the FETCH pass> <kernel family substring> <mode> [detail.txt]
"""
import collections
import csv
import glob
import json
import sys

FETCH_FACTOR = 2.0  # MI355X_MICROARCH.md, HBM: FETCH_SIZE = TCC_EA0_RDREQ x 64 B with 128-B requests tallied at 64 B


def load(pat):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(glob.glob(pat)[0])):
        agg[r["Kernel_Name"]][0] += 1
        agg[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return agg


def shape_groups(detail_path):
    """bench.py --detail rows -> {group: algorithmic bytes per step}; group = the instantiation the shape runs on."""
    g = collections.defaultdict(float)
    for line in open(detail_path).read().splitlines()[1:]:
        v = line.split()
        rows, cin, taps, store = int(v[0]), int(v[1]), int(v[2]), int(v[5])
        alg = float(v[10]) * 1e9
        resblock = rows == cin and rows in (128, 256) and taps in (3, 7, 11)
        if store == 2:
            key = "da<1, 0, 8> polyphase transposed convs"
        elif taps == 1:
            key = "dag / da<0, 0, 8> k = 1"
        elif resblock:
            key = f"da<2, {taps}, *> snake resblock convs, {taps} taps"
        else:
            key = f"da<1, {taps}, *> / other, {taps} taps"
        g[key] += alg
    return g


def kernel_group(name):
    import re
    m = re.search(r"da_kernel<(\d+), (\d+), (\d+)((?:, (?:true|false))*)>", name)
    if m:
        act, kt = int(m.group(1)), int(m.group(2))
        flags = [v.strip() == "true" for v in m.group(4).split(",")[1:]]  # P1, W2, S16, PRE, BF
        pre = len(flags) >= 4 and flags[3]
        if act == 2 and kt in (3, 7, 11):
            return f"da<2, {kt}, *> snake resblock convs, {kt} taps"
        if (act == 1 and kt == 0) or (pre and kt == 0):  # (round 5: the pre-split form of the run-time-tap kernel runs the polyphase convs only)
            return "da<1, 0, 8> polyphase transposed convs"
        if kt == 0:
            return "dag / da<0, 0, 8> k = 1"
        return f"da<1, {kt}, *> / other, {kt} taps"
    if "dag" in name:
        return "dag / da<0, 0, 8> k = 1"
    return "LDS-DMA forms (conv1d_f16x3_kernel<128,..>)"


def main(fetch_dir, write_dir, out, bench_stdout, family="conv1d_f16x3_kernel<128,", mode="f16x3", detail=None):
    f = load(fetch_dir + "/*/*_counter_collection.csv")
    w = load(write_dir + "/*/*_counter_collection.csv")
    fams = family.split("|")  # ("a|b": several name patterns, e.g. the direct-A and the LDS-DMA form of the 128-row conv)
    fam = [k for k in f if any(x in k for x in fams)]  # every instance of the BM=128 family counts as the dominant kernel
    n = sum(f[k][0] for k in fam)
    fetch_kib = sum(f[k][1] for k in fam)
    write_kib = sum(w[k][1] for k in fam)
    bench = json.loads([l for l in open(bench_stdout).read().splitlines() if l.startswith("{")][-1])
    steps = max(int(bench.get("steps", 1)) + int(bench.get("warmup", 0)), 1)  # every step of the run is counted
    st = [k for k in f if "in_stats_kernel" in k]
    check = None
    if st and bench.get("in_stats"):
        got = sum(f[k][1] for k in st) * 1024 * FETCH_FACTOR
        want = bench["in_stats"]["bytes_per_step"] * steps
        check = {"kernel": "in_stats_kernel", "launches": sum(f[k][0] for k in st),
                 "bytes_it_must_read": want, "FETCH_SIZE_x2_bytes": got, "ratio": got / want if want else None,
                 "note": "a CHECK of the x2 rule on a kernel with a known byte count (each tensor read exactly once, 4 B per "
                         "lane coalesced); it is not used to rescale anything"}
    alg = bench["roofline"].get("algorithmic_bytes_per_launch")
    res = {
        "kernel": family + f" (family, {len(fam)} instances)", "launches": n,
        "FETCH_SIZE_KiB_per_launch": fetch_kib / n, "WRITE_SIZE_KiB_per_launch": write_kib / n,
        "fetch_factor": FETCH_FACTOR,
        "fetch_bytes_per_launch": fetch_kib / n * 1024 * FETCH_FACTOR,
        "write_bytes_per_launch": write_kib / n * 1024,
        "x2_rule_check": check,
        "algorithmic_bytes_per_launch": alg,
    }
    res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
    res["traffic_over_algorithmic"] = res["traffic_bytes_per_launch"] / alg if alg else None
    if detail:
        alg_g = shape_groups(detail)
        per = collections.OrderedDict()
        for k in sorted(fam, key=lambda k: -(f[k][1] * FETCH_FACTOR + w[k][1])):
            grp = kernel_group(k)
            e = per.setdefault(grp, {"kernels": [], "launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
            e["kernels"].append(k.split("(")[0][-60:])
            e["launches"] += f[k][0]
            e["fetch_bytes"] += f[k][1] * 1024 * FETCH_FACTOR
            e["write_bytes"] += w[k][1] * 1024
        for grp, e in per.items():
            e["traffic_bytes"] = e["fetch_bytes"] + e["write_bytes"]
            a = alg_g.get(grp, 0.0) * steps
            e["algorithmic_bytes"] = a
            e["traffic_over_algorithmic"] = e["traffic_bytes"] / a if a else None
        res["per_instantiation"] = per
        res["per_instantiation_note"] = ("traffic summed over every launch of the run; algorithmic bytes = bench.py --detail of the same "
                                         "workload, summed over the shapes each instantiation runs (k = 1 shapes that the LDS-DMA "
                                         "statistics form runs are counted under 'dag / da<0, 0, 8> k = 1')")
    sp = [k for k in f if "split_image_kernel" in k]
    if sp:  # the pre-split passes (conv_f16x3_pre.hip): outside the family, reported beside it
        res["split_image_passes"] = {"launches": sum(f[k][0] for k in sp), "fetch_bytes": sum(f[k][1] for k in sp) * 1024 * FETCH_FACTOR,
                                     "write_bytes": sum(w[k][1] for k in sp if k in w) * 1024,
                                     "note": "one elementwise pass per pre-split input image (1024-row decoder convs, polyphase upsamplers): "
                                             "reads the f32 tensor once, writes the hi / lo image once; not counted in the family's traffic"}
    res["workload"] = {"batch": bench["config"]["batch_per_gpu"], "tokens": bench["config"]["tokens"],
                       "frames": bench["config"]["frames"], "conv_mode": mode}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:8])
