"""Sum rocprofv3 counter_collection.csv files of tools/pmc_denoise.sh for the r1d_kernel dispatches."""
import csv, glob, os, sys
root = sys.argv[1]
for grp in sorted(os.listdir(root)):
    files = glob.glob(os.path.join(root, grp, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    tot, n = {}, {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if "r1d_kernel" not in row.get("Kernel_Name", ""):
                continue
            k = row["Counter_Name"]
            tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
            n[k] = n.get(k, set()) | {row["Dispatch_Id"]}
    for k in sorted(tot):
        d = max(1, len(n[k]))
        print(f"{grp:8s} {k:28s} {tot[k] / d:16.0f} per launch ({d} launches)")
