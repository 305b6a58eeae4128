// Sanitizer harness of the CPU-only arithmetic of the host mirror (keaki_amd/host/keaki.cpp): the scalar field (the mirror of what keaki
// takes from ark-ff: Fr::rand, +, -, *, pow, inverse) and the radix-2 domain (ark-poly's Radix2EvaluationDomain: elements, fft, ifft; the
// reference's src/vec.rs:36-37, src/kzg.rs:182-185). Built with -fsanitize=address,undefined by `make host_cpu_asan`; run by
// tests/test_ptau_asan.py. It checks algebraic identities while the sanitizers watch the limb arithmetic; no GPU call is made.
#include <cstdio>
#include <cstdlib>

#include "keaki.hpp"

using namespace keaki;

struct Mix : Rng {
  uint64_t s;
  explicit Mix(uint64_t seed) : s(seed) {}
  uint64_t next_u64() override {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
};
static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main() {
  Mix rng(2026);
  // field identities on random and corner values
  std::vector<Fr> vals = {Fr::zero(), Fr::one(), Fr::from_i64(-1), Fr::from_i64(-24), Fr::from_u64(~0ULL)};
  for (int i = 0; i < 200; i++) vals.push_back(fr_rand(rng));
  for (size_t i = 0; i + 2 < vals.size(); i++) {
    const Fr a = vals[i], b = vals[i + 1], c = vals[i + 2];
    CHECK((a + b) - b == a);
    CHECK(a * (b + c) == a * b + a * c);
    CHECK(-(-a) == a);
    CHECK(a + (-a) == Fr::zero());
    if (!a.is_zero()) CHECK(a * a.inverse() == Fr::one());
    CHECK(a.pow(5) == a * a * a * a * a);
  }
  CHECK(Fr::zero().inverse() == Fr::zero());
  // domains: sizes 1 .. 2^12 and the non-powers of two around them (create() rounds up), fft / ifft round trips, padding with zeros
  const size_t sizes[] = {1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 255, 256, 257, 1000, 4095, 4096};
  for (size_t m : sizes) {
    vec::Radix2Domain d = vec::Radix2Domain::create(m);
    CHECK(d.size >= m && (d.size & (d.size - 1)) == 0 && (d.size == 1 || d.size / 2 < m));
    CHECK(d.group_gen * d.group_gen_inv == Fr::one());
    CHECK(d.size_inv * Fr::from_u64(d.size) == Fr::one());
    std::vector<Fr> el = d.elements();
    CHECK(el.size() == d.size && el[0] == Fr::one());
    if (d.size > 1) CHECK(el[1] == d.group_gen && el[d.size - 1] * d.group_gen == Fr::one());
    std::vector<Fr> v(m);
    for (auto& x : v) x = fr_rand(rng);
    std::vector<Fr> back = d.fft(d.ifft(v));           // evaluations -> coefficients -> evaluations
    CHECK(back.size() == d.size);
    for (size_t i = 0; i < m; i++) CHECK(back[i] == v[i]);
    for (size_t i = m; i < d.size; i++) CHECK(back[i].is_zero());
    std::vector<Fr> co = d.ifft(d.fft(v));             // coefficients -> evaluations -> coefficients
    for (size_t i = 0; i < m; i++) CHECK(co[i] == v[i]);
    // one evaluation against Horner
    if (d.size >= 2) {
      std::vector<Fr> ev = d.fft(v);
      Fr h = Fr::zero();
      for (size_t i = m; i-- > 0;) h = h * el[1] + v[i];
      CHECK(ev[1] == h);
    }
  }
  printf(fails ? "FAILED %d checks\n" : "all checks passed\n", fails);
  return fails ? 1 : 0;
}
