"""Per-kernel effective clock (GRBM_GUI_ACTIVE per XCD / duration) and MFMA-pipe occupancy (SQ_VALU_MFMA_BUSY_CYCLES
per SIMD / active cycles) from one rocprofv3 --kernel-trace --pmc run (tools/pmc_clocks.sh)."""
import collections, csv, re, sys
d = sys.argv[1]
trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(f"{d}/run_kernel_trace.csv"))}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{d}/run_counter_collection.csv")):
    t = trace[r["Dispatch_Id"]]
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:48]
    agg[name][r["Counter_Name"]].append((float(r["Counter_Value"]), int(t["End_Timestamp"]) - int(t["Start_Timestamp"])))
rows = []
for name, v in agg.items():
    g, m = v.get("GRBM_GUI_ACTIVE"), v.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if not g:
        continue
    n = len(g)
    cyc = sum(x[0] for x in g) / n / 8          # per XCD
    dur = sum(x[1] for x in g) / n              # ns
    busy = sum(x[0] for x in m) / n / 1024 if m else 0.0   # per SIMD (256 CUs x 4)
    rows.append((dur * n, name, n, dur / 1e6, cyc / dur, 100 * busy / cyc))
print("| kernel | launches | avg ms | effective clock (GHz) | MFMA pipe busy (% of active cycles) |\n|---|---|---|---|---|")
for _, name, n, ms, clk, busy in sorted(rows, reverse=True)[:14]:
    print(f"| `{name}` | {n} | {ms:.3f} | {clk:.2f} | {busy:.1f} |")
