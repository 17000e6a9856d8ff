"""Sum the SQ counters of tools/pmc_sq.sh for one kernel (all launches of it in the run)."""
import collections, csv, glob, sys
prefix, kern = sys.argv[1], sys.argv[2]
tot = collections.OrderedDict()
n = 0
for d in sorted(glob.glob(prefix + "*/")):
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                    n += 1
print(f"kernel {kern}: {n} launches")
for k, v in tot.items():
    print(f"{k:34s} {v:16.0f}")
