"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) for the dominant kernel.

Units/corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of a coalesced streaming read.  That factor is re-calibrated here
on in_stats_kernel, which reads each activation tensor exactly once with the same 4-byte-per-lane
coalesced pattern as the conv kernel's input staging (known bytes passed on the command line)."""
import collections
import csv
import glob
import json
import sys


def load(pat):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(glob.glob(pat)[0])):
        agg[r["Kernel_Name"]][0] += 1
        agg[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return agg


def main(fetch_dir, write_dir, out, stats_known_bytes_per_step, family="conv1d_mfma_kernel<128, 128, 2, 2>", mode="f32"):
    f = load(fetch_dir + "/*/*_counter_collection.csv")
    w = load(write_dir + "/*/*_counter_collection.csv")
    fam = [k for k in f if family in k]  # every instance of the BM=128 family counts as the dominant kernel
    conv = family + (" (family, %d instances)" % len(fam) if len(fam) > 1 else "")
    f[conv] = [sum(f[k][0] for k in fam), sum(f[k][1] for k in fam)]
    w[conv] = [sum(w[k][0] for k in fam), sum(w[k][1] for k in fam)]
    st = [k for k in f if "in_stats_kernel" in k][0]
    calib = float(stats_known_bytes_per_step) / (f[st][1] * 1024)
    n = f[conv][0]
    res = {
        "kernel": conv, "launches": n,
        "FETCH_SIZE_KiB_per_launch": f[conv][1] / n, "WRITE_SIZE_KiB_per_launch": w[conv][1] / n,
        "fetch_calibration_factor": calib,
        "calibration": f"in_stats_kernel: {f[st][0]} launches, known {float(stats_known_bytes_per_step) / 1e9:.1f} GB read "
                       f"per step vs FETCH_SIZE {f[st][1] * 1024 / 1e9:.1f} GB",
        "fetch_bytes_per_launch_corrected": f[conv][1] / n * 1024 * round(calib),
        "write_bytes_per_launch": w[conv][1] / n * 1024,
    }
    res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch_corrected"] + res["write_bytes_per_launch"]
    res["workload"] = {"batch": 64, "tokens": 130, "frames": 422, "conv_mode": mode}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:7])
