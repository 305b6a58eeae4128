import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as oc
from keaki_amd.hip import KeakiHip
h = KeakiHip(0)
print("selftest mismatches:", h.selftest_field(16, 8, 1))
g1, g2 = oc.generators()
f = oc.miller_loop_raw(g1, g2)
exp = oc.pairing_batch(g1.reshape(1, 8), g2.reshape(1, 16))
got = h.final_exp_batch(np.stack([f, f, f]))
print("final_exp on oracle miller output matches:", [bool(np.array_equal(got[i], exp[0])) for i in range(3)])
gt = h.pairing_batch(np.stack([g1, g1]), np.stack([g2, g2]))
print("full pairing matches:", bool(np.array_equal(gt[0], exp[0])), bool(np.array_equal(gt[1], exp[0])))
fm = h.miller_loop_batch(np.stack([g1, g1]), np.stack([g2, g2]))
e0 = oc.final_exp_raw(fm[0])
expraw = oc.final_exp_raw(f)
print("device miller + oracle final exp matches:", bool(np.array_equal(e0, expraw)), bool(np.array_equal(fm[0], fm[1])))
got2 = h.final_exp_batch(fm)
print("device miller + device final exp (separate kernels) matches:", bool(np.array_equal(got2[0], exp[0])))
import bn254_py as py, importlib.util
spec = importlib.util.spec_from_file_location("gc", os.path.join(ROOT, "keaki_amd", "csrc", "gen_constants.py")); gc = importlib.util.module_from_spec(spec); spec.loader.exec_module(gc)
naf = gc.naf_6z2()
# expected line sequence with the device's NAF (same formulas as the oracle's _line_double/_line_add)
q = py.G2_GEN
r = (q[0], q[1], py.F2_ONE); exp_lines = []
negq = py.g2_neg(q)
for i in range(len(naf) - 2, -1, -1):
    r, l = py._line_double(r); exp_lines.append(l)
    if naf[i] == 1: r, l = py._line_add(r, q); exp_lines.append(l)
    elif naf[i] == -1: r, l = py._line_add(r, negq); exp_lines.append(l)
q1 = py._mul_by_char(q); q2 = py._mul_by_char(q1); q2 = (q2[0], py.f2_neg(q2[1]))
r, l = py._line_add(r, q1); exp_lines.append(l)
r, l = py._line_add(r, q2); exp_lines.append(l)
tab = h.g2_prepare(g2)
def dev_line(li):
    out = []
    for c in range(3):
        comp = [oc.limbs_to_ints(oc.fq_from_mont(tab[li, par, c].reshape(1, 4)))[0] for par in (0, 1)]
        out.append(tuple(comp))
    return tuple(out)
bad = [li for li in range(len(exp_lines)) if dev_line(li) != exp_lines[li]]
print("lines:", len(exp_lines), "first mismatching line index:", bad[:5])
if bad:
    li = bad[0]; print("dev", dev_line(li)); print("exp", exp_lines[li])
