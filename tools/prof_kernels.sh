#!/bin/bash
# rocprofv3 kernel summary of a python tool.  Run on the GPU box from the repo root:
#   bash tools/prof_kernels.sh <tag> tools/probe_encoder.py [args]   -> gpurun_out/prof_<tag>/ + a top-N table on stdout
tag=$1; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o run -- /usr/bin/python3 $root/"$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
python3 - $tag <<'PY'
import csv, glob, sys
fs = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    print(f'{r["Name"][:84]:84s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e6:8.3f} ms  total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms  {float(r["Percentage"]):5.1f} %')
PY
