"""Timeline of ONE forward from a rocprofv3 --kernel-trace CSV (batch-1 latency work): every kernel in start order with its stream,
start offset, duration and the idle gap before it (time during which NO kernel of the process was running), then the totals.
    python tools/b1_timeline.py <dir with *_kernel_trace.csv> [which forward, default -2 = the last but one] [out.txt]
A forward = from a style_fc_kernel to the next istft_ola_kernel."""
import csv, glob, os, sys

def main(d, which=-2, out=None):
    f = sorted(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")) + glob.glob(os.path.join(d, "*_kernel_trace.csv")))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "style_fc_kernel" in r["Kernel_Name"]]
    s0 = starts[which]
    e0 = next(i for i in range(s0, len(rows)) if "istft_ola_kernel" in rows[i]["Kernel_Name"])
    fw = rows[s0:e0 + 1]
    t0 = int(fw[0]["Start_Timestamp"])
    lines = [f"# one forward of {os.path.basename(os.path.normpath(d))}: {len(fw)} kernels", f"{'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s} {'strm':>4s}  kernel"]
    busy_end, idle, busy = t0, 0.0, 0.0
    sync_gap = None
    by_name = {}
    for r in fw:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = max(0, s - busy_end) / 1e3
        idle += gap
        busy += max(0, e - max(s, busy_end)) / 1e3
        name = r["Kernel_Name"].replace("void kx::", "").replace("kx::", "").split("(")[0][:70]
        by_name.setdefault(name, [0, 0.0])
        by_name[name][0] += 1
        by_name[name][1] += (e - s) / 1e3
        lines.append(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {r.get('Stream_Id', '?'):>4s}  {name}")
        if "duration_kernel" in r["Kernel_Name"]:
            sync_gap = ("after", e)
        elif sync_gap and sync_gap[0] == "after" and "__amd_rocclr" not in r["Kernel_Name"]:
            sync_gap = ("gap", (s - sync_gap[1]) / 1e3)
        busy_end = max(busy_end, e)
    span = (busy_end - t0) / 1e3
    lines += ["", f"# span {span:.1f} us; some kernel running {busy:.1f} us; nothing running {idle:.1f} us ({100 * idle / span:.1f} %)",
              f"# gap behind duration_kernel (the forward's one host round trip: D2H of the frame counts, sizing, first back-half launch): "
              f"{sync_gap[1] if sync_gap and sync_gap[0] == 'gap' else float('nan'):.1f} us", "# kernel time by name (sum over streams; overlapping kernels both count):"]
    for n, (c, t) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:25]:
        lines.append(f"#   {t:8.1f} us {c:4d} x  {n}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print("\n".join(lines[-32:]))

if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -2, sys.argv[3] if len(sys.argv) > 3 else None)
