"""Prints the kernel timeline of the LAST MSM in a rocprofv3 kernel trace (csv): start offset, duration, kernel name.
    python bench_tools/kernel_timeline.py gpurun_out/sm/sm_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "k_part_count"
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
start = idx[-1] if idx else max(0, len(rows) - 30)
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    print("%8.1f us +%7.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][-48:]))
