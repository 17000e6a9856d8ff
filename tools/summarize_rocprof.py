"""Turn a rocprofv3 --kernel-trace --stats CSV pair into the short summary kept under profiles/."""
import csv
import glob
import json
import os
import sys


def main(prof_dir, bench_json, out, cmd=""):
    stats = glob.glob(os.path.join(prof_dir, "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(prof_dir)})",
             "# command: rocprofv3 --kernel-trace --stats --output-format csv -- " + (cmd or "python3 bench.py --steps 3 --warmup 1 "
                                                                                      "--cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0"),
             f"# total GPU kernel time {tot / 1e6:.2f} ms", "",
             f"{'kernel':88s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>7s}"]
    for r in rows:
        if float(r["Percentage"]) < 0.005:
            continue
        lines.append(f"{r['Name'][:88]:88s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:10.2f} "
                     f"{float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}%")
    if bench_json and os.path.exists(bench_json):
        d = json.loads(open(bench_json).read().strip().splitlines()[-1])
        r = d["roofline"]
        lines += ["", "# bench.py line of the same run (HIP events inside the library, timed steps only):",
                  f"#   value {d['value']:.1f} {d['unit']}, {d['ms_per_step']:.1f} ms/step, {d['utterances_per_s']:.1f} utt/s",
                  f"#   roofline kernel {r['kernel']}",
                  f"#   avg launch {r['avg_launch_ms'] * 1e3:.1f} us over {r['launches_per_step']:.0f} launches/step, "
                  f"{r['gflop_per_launch']:.1f} GFLOP/launch -> {r['achieved']:.1f} TFLOP/s = {r['frac'] * 100:.1f}% of "
                  f"{r['peak']} TFLOP/s (dense MFMA peak of the kernel's dtype)"]
        fam = [x for x in rows if "conv1d_mfma_kernel<128, 128, 2, 2>" in x["Name"]
               or "conv1d_f16x3_kernel<128," in x["Name"] or "conv1d_f16x3_da_kernel<" in x["Name"] or "conv1d_f16x3_dag_kernel<" in x["Name"]]
        if fam:
            calls = sum(int(x["Calls"]) for x in fam)
            tot_ns = sum(float(x["TotalDurationNs"]) for x in fam)
            lines.append(f"#   rocprof average over the same kernel family: {tot_ns / calls / 1e3:.1f} us "
                         f"({calls} calls incl. warm-up)")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
