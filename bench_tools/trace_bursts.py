import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:64]))
ev.sort()
bursts, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur) > 50e6: bursts.append(cur); cur = [e]
    else: cur.append(e)
bursts.append(cur)
for b in bursts[-4:]:
    t0 = b[0][0]
    print("burst: %d kernels, span %.3f ms" % (len(b), (max(e[1] for e in b) - t0) / 1e6))
    for e in b:
        if e[1] - e[0] > 20e3: print("  %7.3f %7.3f  %s" % ((e[0] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2]))
