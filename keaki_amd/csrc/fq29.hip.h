// Fq-typed entry points of the 9 x 29-bit lazy arithmetic; the arithmetic itself is in fq29_core.hip.h (see its header).
#pragma once
#include "bn254_field.hip.h"

namespace bn254 {

KDEV Fq u29_to_fq(const U29& a) {
  Fq r;
  u29_to_sat(r.l, a);
  return r;
}
// saturated canonical residue -> lazy 2^261-form below 2p
KDEV U29 u29_from_fq(const Fq& x) { return u29_mul(u29_from_sat_shift5(x.l), u29_one()); }
// exact zero test (mod p) of a lazy value below ~100p
KDEV bool u29_is_zero(const U29& x) {
  U29 t = u29_mul(x, u29_one());    // same residue, < 2p, limbs exact: zero is 0 or p
  u32 z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) { z |= t.l[i]; e |= t.l[i] ^ Q29::MOD[i]; }
  return z == 0 || e == 0;
}

}  // namespace bn254
