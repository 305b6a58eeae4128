"""One launch each of the pairing kernels over the same 256 pairs (lines on the fly): lane-pair k_pairing, twelve-lane k_pairing_wide (one wave),
k_pairing_wide2<false> (line wave + f wave); and the tabulated-lines form through the GT table of a new commitment (k_pairing_wide2<true>, 260 items).
Run under   rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d <dir> -- python3 bench_tools/profile_pairing_wide.py
and summarise with   python3 bench_tools/profile_pairing_wide.py --summarise <dir>   (VALU instructions per wave and per pairing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    import csv, glob, collections
    agg = collections.OrderedDict()
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "pairing" not in k: continue
            agg.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    items = {"bn254::k_pairing": 32, "bn254::pw::k_pairing_wide": 4, "void bn254::pw::k_pairing_wide2<false>": 4, "void bn254::pw::k_pairing_wide2<true>": 4}
    for k, c in agg.items():
        valu, waves = c["SQ_INSTS_VALU"], c["SQ_WAVES"]
        per_wave = sum(valu) / sum(waves)
        per_wg = per_wave * (2 if "wide2" in k else 1)
        print("%-46s launches %d  waves/launch %.0f  VALU per wave %.0f  per workgroup %.0f  per pairing %.0f (wave-instructions)"
              % (k, len(valu), sum(waves) / len(waves), per_wave, per_wg, per_wg / items.get(k, 1)))
    sys.exit(0)
import numpy as np
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs
from oracle import bn254_py as py
limbs = lambda x: np.frombuffer(int(x).to_bytes(32, "little"), np.uint64)
R256 = 1 << 256
g1 = np.concatenate([limbs(py.G1_GEN[0] * R256 % py.P), limbs(py.G1_GEN[1] * R256 % py.P)])
g2 = np.concatenate([limbs(c * R256 % py.P) for c in (py.G2_GEN[0][0], py.G2_GEN[0][1], py.G2_GEN[1][0], py.G2_GEN[1][1])])
h = KeakiHip(0)
n = 256
P = h.g1_mul_batch(g1, random_fr_limbs(n, 11)); Q = h.g2_mul_batch(g2, random_fr_limbs(n, 12))
h.set_option("pair_wide_max", 0); a = h.pairing_batch(P, Q)
h.set_option("pair_wide_max", 1 << 20); h.set_option("pair_two_waves", 0); b = h.pairing_batch(P, Q)
h.set_option("pair_two_waves", 1); c = h.pairing_batch(P, Q)
assert np.array_equal(a, b) and np.array_equal(a, c)
tau = h.g2_mul_batch(g2, random_fr_limbs(1, 6))[0]
A, V, Rr = random_fr_limbs(8, 1), random_fr_limbs(8, 2), random_fr_limbs(8, 3)
h.encap_batch(P[0], tau, A, V, Rr, 32)          # first call of the context: B's table (a k_pairing_wide2<true> launch of its own)
h.encap_batch(P[1], tau, A, V, Rr, 32)          # a new commitment: 260 pairings with tabulated lines
print("ok")
