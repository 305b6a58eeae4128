"""CPU checks of the pairing tower's round-2 restructuring against the big-int oracle (no GPU): the generated final-exponentiation
PROGRAM (keaki_amd/csrc/pair261_constants.hip.h: FE_PROG, an accumulator machine over slots) computes arkworks' final exponentiation;
the Miller step list follows the NAF of 6z + 2; the written-out Fq6 / sparse-line / Fq4-squaring formulas of pair261.hip.h (sums of
up to three Fq2 products per output coefficient) equal the oracle's tower arithmetic; and the limb-level helpers (multiplication by
9 + u on 29-bit limbs, column budgets of the six-product stream) hold their stated bounds."""
import os
import random
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _consts():
    t = open(os.path.join(ROOT, "keaki_amd", "csrc", "pair261_constants.hip.h")).read()
    arr = lambda name: [int(x) for x in re.search(name + r"\[\d+\] = \{([^}]*)\}", t).group(1).split(",")]
    return arr("MILLER_STEPS"), arr("FE_PROG"), int(re.search(r"FE_EASY_OPS = (\d+)", t).group(1))


def test_generated_header_is_current():
    import subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "keaki_amd", "csrc", "gen_constants.py"), "--pair261"], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "keaki_amd", "csrc", "pair261_constants.hip.h")).read()


def rand_f12(py, rng):
    r2 = lambda: (rng.randrange(py.P), rng.randrange(py.P))
    return ((r2(), r2(), r2()), (r2(), r2(), r2()))


def test_final_exponentiation_program_equals_oracle(py):
    _, prog, n_easy = _consts()
    rng = random.Random(5)
    for _ in range(2):
        f = rand_f12(py, rng)
        slots, acc = {}, f
        for pc, b in enumerate(prog):
            code, s = b & 15, b >> 4
            if code == 0: acc = slots[s]
            elif code == 1: slots[s] = acc
            elif code == 2:
                acc = py.f12_sqr(acc)                  # the cyclotomic squaring IS a squaring (valid after the easy part)
                assert pc >= n_easy
            elif code == 3: acc = py.f12_mul(acc, slots[s])
            elif code == 4: acc = py.f12_mul(acc, py.f12_conj(slots[s]))
            elif code == 5: acc = py.f12_conj(acc)
            elif code == 6: acc = py.f12_frob(acc, s)
            elif code == 7: acc = py.f12_inv(acc)
            else: raise AssertionError(code)
        assert acc == py.final_exponentiation(f)
        nslots = int(re.search(r"FE_NSLOTS = (\d+)", open(os.path.join(ROOT, "keaki_amd", "csrc", "pair261_constants.hip.h")).read()).group(1))
        assert max(slots) < nslots <= 16                # the slot index travels in the high four bits of an op


def test_miller_steps_follow_the_naf(py):
    steps, _, _ = _consts()
    # replay the scalar: doublings and additions as the list says must end at 6z + 2
    assert steps[0] == 0 and steps[-2:] == [4, 5]
    acc = 1                                         # T = Q: the top digit of the NAF
    for st in steps[:-2]:
        if st in (0, 1): acc *= 2
        elif st == 2: acc += 1
        elif st == 3: acc -= 1
        else: raise AssertionError(st)
    assert acc == 6 * py.Z + 2
    assert steps.count(0) == 1 and len(steps) <= 96


def test_written_out_tower_formulas(py):
    rng = random.Random(7)
    r2 = lambda: (rng.randrange(py.P), rng.randrange(py.P))
    xi = py.f2_mul_xi
    m, add = py.f2_mul, py.f2_add
    for _ in range(5):
        a, b = (r2(), r2(), r2()), (r2(), r2(), r2())
        c0 = add(add(m(a[0], b[0]), m(xi(a[1]), b[2])), m(xi(a[2]), b[1]))
        c1 = add(add(m(a[0], b[1]), m(a[1], b[0])), m(xi(a[2]), b[2]))
        c2 = add(add(m(a[0], b[2]), m(a[1], b[1])), m(a[2], b[0]))
        assert (c0, c1, c2) == py.f6_mul(a, b)
        # sparse line product f * (c0 + (d0 + d1 v) w)
        f = rand_f12(py, rng)
        lc0, d0, d1 = r2(), r2(), r2()
        (a0, a1, a2), (e0, e1, e2) = f
        s3 = lambda x, y, z: add(add(x, y), z)
        r0 = (s3(m(a0, lc0), m(e1, xi(d1)), m(e2, xi(d0))), s3(m(a1, lc0), m(e0, d0), m(e2, xi(d1))), s3(m(a2, lc0), m(e0, d1), m(e1, d0)))
        r1 = (s3(m(e0, lc0), m(a0, d0), m(a2, xi(d1))), s3(m(e1, lc0), m(a0, d1), m(a1, d0)), s3(m(e2, lc0), m(a1, d1), m(a2, d0)))
        assert (r0, r1) == py.f12_mul_by_034(f, lc0, d0, d1)
        # Fq4 squaring (x + y s)^2, s^2 = xi
        x, y = r2(), r2()
        t0 = add(m(x, x), m(xi(y), y)); t1 = py.f2_dbl(m(x, y))
        assert t0 == add(py.f2_sqr(x), xi(py.f2_sqr(y))) and t1 == m(py.f2_dbl(x), y)


def test_limb_helpers_and_column_budget(py):
    """xi_limbs (pair261.hip.h) on 29-bit limbs: value (9 a -/+ o + K p), every intermediate below 2^32, output limbs <= 2^29 + 8;
    the six-product stream's column sums stay below 2^64 at the limb bound."""
    P, MASK = py.P, (1 << 29) - 1
    limbs = lambda v: [(v >> (29 * i)) & MASK for i in range(8)] + [v >> 232]
    val = lambda l: sum(x << (29 * i) for i, x in enumerate(l))
    def biased(c, bias_log=30):
        k = limbs(c * P); b, cy = 1 << bias_log, (1 << bias_log) >> 29
        return [k[0] + b] + [k[i] + b - cy for i in range(1, 8)] + [k[8] - cy]
    def carry(x):
        r = [x[0] & MASK] + [(x[i] & MASK) + (x[i - 1] >> 29) for i in range(1, 8)] + [x[8] + (x[7] >> 29)]
        return r
    def xi_limbs(a, o, K, odd):
        t = []
        for i in range(9):
            addv = o[i] if odd else K[i] - o[i]
            assert 0 <= addv < 1 << 32
            lo8 = ((a[i] << 3) & MASK) if i < 8 else (a[i] << 3)
            v = lo8 + a[i] + addv + ((a[i - 1] >> 26) if i else 0)
            assert v < 1 << 32
            t.append(v)
        return carry(t)
    rng = random.Random(11)
    for bound, kmul in ((1, 2), (2, 4)):
        K = biased(kmul)
        for trial in range(300):
            a = rng.randrange(bound * P) if trial else bound * P - 1
            o = rng.randrange(bound * P) if trial > 1 else bound * P - 1
            for odd in (False, True):
                r = xi_limbs(limbs(a), limbs(o), K, odd)
                assert all(x <= (1 << 29) + 8 for x in r[:8]) and r[8] < 1 << 29
                exp = 9 * a + o if odd else 9 * a - o + kmul * P
                assert val(r) == exp and val(r) < (9 * bound + kmul) * P
    lim = (1 << 29) + 8
    worst = 6 * 9 * lim * lim + 9 * MASK * MASK
    assert worst + (worst >> 29) < 1 << 64
    # dual stream with one side doubled (fq4_sqr: t1 = (2x) y): a, c < 2^30, b, d <= 2^29 + 8
    assert 2 * 9 * (1 << 30) * lim + 9 * MASK * MASK < 1 << 64
    # three-product stream with ONE wide operand (fq4_sqr: t0 = sx sy + (xi y) y, sx = x0 + x1 with limbs below 2^30): 9 wide products,
    # 18 ordinary ones, 9 reduction products and the carry of the column below
    worst3 = 9 * (1 << 30) * lim + 18 * lim * lim + 9 * MASK * MASK
    assert worst3 + (worst3 >> 29) < 1 << 64
    # its value: even lane (x0 + x1)(x0 - x1 + 2p) < 2p * 3p, the xi products 11p * p + 11p * 2p: (6 + 33) p^2 / 2^261 + p < 2p
    assert (6 + 33) * P * P // (1 << 261) + P < 2 * P


def test_g2_lazy_mixed_addition_model():
    """keaki_amd/csrc/models/model_g2_add29.py: the G2 mixed addition of xyzz29_g2.hip.h operation by operation on Python integers, with
    assertions on every limb (no negative value, no overflow, stream budgets) and every stated bound, against plain Fq2 XYZZ arithmetic"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("model_g2_add29", os.path.join(ROOT, "keaki_amd", "csrc", "models", "model_g2_add29.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    maxima = m.run(trials=120, seed=3)
    assert maxima["X3"] < 1.1 and maxima["Yacc"] < 1.5 and maxima["P"] < 5.3 and maxima["R"] < 7.3


def test_jacobian_ladder_formulas_model():
    """keaki_amd/csrc/models/model_jac29.py: the point arithmetic of jac29.hip.h (FK23 ladders) operation by operation on Python integers --
    the doubling and the mixed addition with Y3 as ONE dual stream each, the butterflies' shared add / subtract pair, the effective-affine
    window tables in both shapes, the pair of tables and the two-term ladders of the radix-4 passes -- with assertions on every limb and every
    value bound; whole ladders against plain affine arithmetic."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("model_jac29", os.path.join(ROOT, "keaki_amd", "csrc", "models", "model_jac29.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    maxima = m.run(seed=5, ladders=4)
    assert maxima["dbl.x"] < 17.6 and maxima["dbl.y"] < 3.8 and maxima["dbl.z"] < 2.1 and maxima["madd.y"] < 1.8 and maxima["madd.z"] < 1.5
    assert maxima["tab.y"] < 2.0 and maxima["pair.y"] < 2.0 and maxima["as.y"] < 1.45
    # worst cases by the stream rule value < (sum of products) / 2^261 + p, 2^261 / p > 169:
    #   dbl: A < 19^2/169 + 1 = 3.14, E = 3A < 9.5, T < 4 X B + 32 < 37.5, B < 3.2 (Y < 19), 2B (16 - 4B) <= 6.4 * 16
    assert (9.5 * 37.5 + 6.4 * 16) / 169 + 1 < 3.8
    #   madd: H < 2 + 32 = 34, HH < 34^2/169 + 1 = 7.9, H^3 < 34 * 7.9/169 + 1 = 2.6, V < 19 * 7.9/169 + 1 = 1.9, r < 2 + 4 = 6, T < V + 16 = 17.9, 8p - Y1 <= 8
    assert (6 * 17.9 + 8 * 2.6) / 169 + 1 < 1.8


def test_division_step_inverse_model():
    """keaki_amd/csrc/models/model_safegcd.py: the constant-time division-step inverse of bn254_field.hip.h (20 x 30 steps on nine signed 30-bit
    limbs) statement by statement, with the exactness of every shift asserted, against pow(x, -1, p) on random and boundary values"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("model_safegcd", os.path.join(ROOT, "keaki_amd", "csrc", "models", "model_safegcd.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    assert m.run(n=120, seed=9)


def test_twelve_lane_kernel_flat_basis_formulas(py):
    """pairing_wide.hip.h, in integers: pair k of a row holds the coefficient of w^k (Fq12 = Fq2[w]/(w^6 - xi), tower coefficient c_e.c_j at
    k = 2j + e). The schoolbook product with xi where the index wraps, the sparse line product (c0 at w^0, d0 at w^1, d1 at w^3), the
    Granger-Scott squaring by pairs (a0,a3), (a1,a4), (a2,a5) with the signs of the kernel, the replicated-T line rounds, and the value bounds
    of the streams as the header states them."""
    rng = random.Random(11)
    r2 = lambda: (rng.randrange(py.P), rng.randrange(py.P))
    xi, m, add, sub, dbl = py.f2_mul_xi, py.f2_mul, py.f2_add, py.f2_sub, py.f2_dbl
    flat = lambda f: [f[k & 1][k >> 1] for k in range(6)]
    tower = lambda a: ((a[0], a[2], a[4]), (a[1], a[3], a[5]))
    zero = (0, 0)
    for _ in range(3):
        f, g = rand_f12(py, rng), rand_f12(py, rng)
        a, b = flat(f), flat(g)
        c = []
        for k in range(6):
            acc = zero
            for i in range(6):
                wrap = k < i
                j = k + 6 - i if wrap else k - i
                acc = add(acc, m(a[i], xi(b[j]) if wrap else b[j]))
            c.append(acc)
        assert tower(c) == py.f12_mul(f, g)
        # line product
        lc0, d0, d1 = r2(), r2(), r2()
        c = []
        for k in range(6):
            i1, i3 = (k - 1 if k >= 1 else 5), (k - 3 if k >= 3 else k + 3)
            c.append(add(add(m(a[k], lc0), m(a[i1], d0 if k >= 1 else xi(d0))), m(a[i3], d1 if k >= 3 else xi(d1))))
        assert tower(c) == py.f12_mul_by_034(f, lc0, d0, d1)
        # memory position of w^k in the tower's (and ark-serialize's) order c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2
        mem = [f[0][0], f[0][1], f[0][2], f[1][0], f[1][1], f[1][2]]
        assert all(mem[(k & 1) * 3 + (k >> 1)] == a[k] for k in range(6))
        # Frobenius: coefficient-wise (the constants are indexed by the power of w)
        assert tower([py.f2_mul(py.f2_conj(a[k]), py._FROB_W[1][k]) for k in range(6)]) == py.f12_frob(f, 1)
        assert tower([py.f2_mul(a[k], py._FROB_W[2][k]) for k in range(6)]) == py.f12_frob(f, 2)
        # conjugation f^(p^6): the odd powers of w change sign
        assert tower([py.f2_neg(a[k]) if k & 1 else a[k] for k in range(6)]) == py.f12_conj(f)
    # cyclotomic squaring: only valid on the cyclotomic subgroup -- take an element of it (easy part of the final exponentiation)
    f = rand_f12(py, rng)
    u = py.f12_mul(py.f12_conj(f), py.f12_inv(f))
    u = py.f12_mul(py.f12_frob(u, 2), u)
    a = flat(u)
    out = []
    for k in range(6):
        xs = 0 if k in (0, 3) else (1 if k in (2, 5) else 2)
        x, y = a[xs], a[xs + 3]
        te = add(m(x, x), m(xi(y), y)); to = dbl(m(x, y))
        t = te if k % 2 == 0 else (xi(to) if k == 1 else to)
        d = sub(t, a[k]) if k % 2 == 0 else add(t, a[k])
        out.append(add(dbl(d), t))
    assert tower(out) == py.f12_sqr(u)
    # bounds (multiples of p, over 169 = 2^261 / p): general product 3 per plain term, 27 per wrapped term, at most three wrapped terms per stream;
    # line product c0 4 + xi d0 54 + xi d1 27; cyclotomic 6 + 33
    plain, wrapped = 1 * 1 + 1 * 2, 1 * 11 + 1 * 16
    assert 3 * wrapped <= 169 and plain + 2 * wrapped <= 169 and 4 + (22 + 32) + (11 + 16) <= 169 and 6 + 33 <= 169
