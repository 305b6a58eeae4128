"""Folds the counter passes of bench_tools/r6_fetch_calibration.sh: counter value per kernel of ubench_fetch_calib / bytes the kernel requested."""
import csv, glob, os, re, sys
d = sys.argv[1]
req = {}
for line in open(os.path.join(d, "plain_run.txt")):
    m = re.match(r"(\w+)\s+requested_bytes (\d+)\s+([0-9.]+) ms\s+([0-9.]+) TB/s", line)
    if m:
        req[m.group(1)] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
names = {"k_calib_stream16": "stream16", "k_calib_stream4": "stream4", "k_calib_row64<0>": "row64", "k_calib_row64<1>": "row64_nt",
         "k_calib_row64_coop<0>": "row64_coop", "k_calib_row64_coop<1>": "row64_coop_nt", "k_calib_row128": "row128", "k_calib_row64_pairs": "row64_pairs", "k_calib_halves": "halves_2MiB_1wg"}
vals = {}
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].strip()
        k = re.sub(r"^void ", "", k)
        if k in names:
            vals.setdefault(names[k], {})[row["Counter_Name"]] = float(row["Counter_Value"])
print("# table: %.0f MiB, every byte requested exactly once by every kernel; FETCH_SIZE / WRITE_SIZE are reported in KB (x 1024 below)" % (next(iter(req.values()))[0] / 2**20))
print("%-14s %9s %8s | %-22s | %s" % ("shape", "ms", "TB/s", "FETCH_SIZE / requested", "other counters per 64 B requested"))
for shape, (b, ms, tbs) in req.items():
    v = vals.get(shape, {})
    f = v.get("FETCH_SIZE")
    others = "  ".join("%s %.3f" % (c.replace("_sum", ""), x / (b / 64.0)) for c, x in sorted(v.items()) if c not in ("FETCH_SIZE", "WRITE_SIZE"))
    print("%-14s %9.3f %8.2f | %-22s | %s" % (shape, ms, tbs, ("%.4f" % (f * 1024.0 / b)) if f is not None else "n/a", others))
