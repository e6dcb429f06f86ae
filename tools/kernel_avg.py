"""Average duration per launch of the kernels whose name contains one of the given substrings, from a rocprofv3
`*_kernel_stats.csv` found under a directory.    python tools/kernel_avg.py DIR substr [substr ...]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
for sub in sys.argv[2:]:
    sel = [r for r in rows if sub in r["Name"]]
    calls = sum(int(r["Calls"]) for r in sel)
    tot = sum(float(r["TotalDurationNs"]) for r in sel)
    print(f"{sub:28s} calls {calls:6d}  total {tot / 1e6:9.3f} ms  avg {tot / max(calls, 1) / 1e3:8.2f} us")
