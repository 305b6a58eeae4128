import os, sys, time
sys.path.insert(0, os.getcwd())
from keaki_amd.hip import KeakiHip
hip = KeakiHip(0)
t0 = time.time(); tot = 0
for seed in range(100, 140):
    b = hip.selftest_field(4096, 256, seed); tot += b
    if b: print("seed", seed, "mismatches", b)
print("selftest_field 40 seeds x 4096 blocks x 256 iters: mismatches", tot, "%.1fs" % (time.time() - t0))
