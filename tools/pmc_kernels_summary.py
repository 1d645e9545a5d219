"""Per-kernel summary of a tools/pmc_kernels.sh collection: average duration from the --stats run, counters per launch
(mean over the launches) from the --pmc runs.  `python tools/pmc_kernels_summary.py gpurun_out/pmck_<tag> [out.json]`.
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; both the raw bytes and the bytes with the gfx950 doubling of
FETCH_SIZE (exact for 16-B/lane streaming reads only: MI355X_MICROARCH.md, HBM) are written."""
import csv, glob, json, os, re, sys
root = sys.argv[1]
kMinClockUs = 20.0   # shortest launch for which GRBM_GUI_ACTIVE / duration is read as a clock


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:100]


kern = {}
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = kern.setdefault(short(row["Name"]), {})
        k["calls"] = int(row["Calls"])
        k["avg_us"] = float(row["AverageNs"]) / 1e3
        k["min_us"] = float(row["MinNs"]) / 1e3
        k["total_pct"] = float(row["Percentage"])
for grp in ("fetch", "write", "mfma", "wait", "lds", "l2"):
    tot, n = {}, {}
    for f in glob.glob(os.path.join(root, grp, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            key = (short(row["Kernel_Name"]), row["Counter_Name"])
            tot[key] = tot.get(key, 0.0) + float(row["Counter_Value"])
            n.setdefault(key, set()).add(row["Dispatch_Id"])
    for (kn, cn), v in tot.items():
        kern.setdefault(kn, {})[cn] = v / max(1, len(n[(kn, cn)]))
for kn, k in kern.items():
    if "FETCH_SIZE" in k:
        k["fetch_bytes_raw"] = int(k["FETCH_SIZE"] * 1024)
        k["fetch_bytes_corrected"] = int(k["FETCH_SIZE"] * 1024 * 2)
    if "WRITE_SIZE" in k:
        k["write_bytes"] = int(k["WRITE_SIZE"] * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "GRBM_GUI_ACTIVE" in k and k["GRBM_GUI_ACTIVE"] > 0:
        # busy cycles summed over 1024 SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs
        k["mfma_busy_frac"] = (k["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (k["GRBM_GUI_ACTIVE"] / 8.0)
    if "GRBM_GUI_ACTIVE" in k and k.get("avg_us", 0) >= kMinClockUs:
        # active cycles of an XCD / duration: a clock only for launches much longer than what GRBM_GUI_ACTIVE also counts
        # around a dispatch (command processor, cache invalidation: several us) -- a 5 us kernel "ran at 5.5 GHz" this way.
        # Below kMinClockUs the field is left out and the table prints n/a.
        k["effective_clock_ghz"] = k["GRBM_GUI_ACTIVE"] / 8.0 / k["avg_us"] / 1e3
    if "SQ_WAVE_CYCLES" in k and "SQ_WAIT_ANY" in k and k["SQ_WAVE_CYCLES"] > 0:
        # the three are disjoint shares of the wave cycles (MI355X_MICROARCH.md, SQ counters)
        k["wait_any_frac"] = k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"]
        k["wait_inst_frac"] = k.get("SQ_WAIT_INST_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
        k["active_inst_frac"] = k.get("SQ_ACTIVE_INST_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
    if "SQ_LDS_IDX_ACTIVE" in k and k["SQ_LDS_IDX_ACTIVE"] > 0:
        k["lds_conflict_frac"] = k.get("SQ_LDS_BANK_CONFLICT", 0.0) / k["SQ_LDS_IDX_ACTIVE"]
        if k.get("GRBM_GUI_ACTIVE", 0) > 0:   # LDS-array cycles summed over 256 CUs against the kernel's cycles
            k["lds_busy_frac"] = (k["SQ_LDS_IDX_ACTIVE"] / 256.0) / (k["GRBM_GUI_ACTIVE"] / 8.0)
    if "TCC_HIT_sum" in k and k["TCC_HIT_sum"] + k.get("TCC_MISS_sum", 0.0) > 0:
        k["l2_hit_frac"] = k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"])
rows = sorted(kern.items(), key=lambda kv: -kv[1].get("total_pct", 0.0))
for kn, k in rows:
    if "avg_us" not in k:
        continue
    print(f"{kn:70s} calls {k['calls']:4d} avg {k['avg_us']:10.1f} us  {k['total_pct']:5.1f} %  fetch(raw) "
          f"{k.get('fetch_bytes_raw', 0) / 1e6:9.1f} MB write {k.get('write_bytes', 0) / 1e6:9.1f} MB  mfma busy "
          f"{k.get('mfma_busy_frac', float('nan')):.3f} clock "
          + (f"{k['effective_clock_ghz']:.2f} GHz" if "effective_clock_ghz" in k else f"n/a (< {kMinClockUs:.0f} us)")
          + (f"  wait {k['wait_any_frac']:.2f} stall {k['wait_inst_frac']:.2f} issue {k['active_inst_frac']:.2f}" if "wait_any_frac" in k else "")
          + (f"  lds busy {k.get('lds_busy_frac', float('nan')):.2f} conflict {k['lds_conflict_frac']:.2f}" if "lds_conflict_frac" in k else "")
          + (f"  l2 hit {k['l2_hit_frac']:.2f}" if "l2_hit_frac" in k else ""))
if len(sys.argv) > 2:
    json.dump(dict(note="tools/pmc_kernels.sh: per-kernel means per launch; one counter group per rocprofv3 run", kernels=dict(rows)),
              open(sys.argv[2], "w"), indent=1)
