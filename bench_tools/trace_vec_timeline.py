"""Timeline (ms from the first event of the last burst) of kernels and copies from a rocprofv3 csv directory."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", ""), r.get("Bytes", r.get("Size", "")))))
ev.sort()
# bursts: gaps > 30 ms
bursts, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur) > 30e6: bursts.append(cur); cur = [e]
    else: cur.append(e)
bursts.append(cur)
b = bursts[-1]
t0 = b[0][0]
busy_k = sum(e[1] - e[0] for e in b if e[2][0] == "K"); busy_c = sum(e[1] - e[0] for e in b if e[2][0] == "C")
print("last burst: %d events, span %.2f ms, kernels %.2f ms, copies %.2f ms" % (len(b), (max(e[1] for e in b) - t0) / 1e6, busy_k / 1e6, busy_c / 1e6))
for e in b:
    if e[1] - e[0] > 50e3: print("%8.2f %8.2f  %s" % ((e[0] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2]))
