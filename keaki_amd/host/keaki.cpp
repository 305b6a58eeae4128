// Host mirror of keaki's public API (see keaki.hpp). Group arithmetic = C-ABI calls into libkeaki_hip.so.
#include "keaki.hpp"

#include <cstdlib>
#include <cstring>
#include <thread>
#include <sys/mman.h>

namespace keaki {

// ------------------------------------------------------------------------------------------------ Fr
namespace {
typedef unsigned __int128 u128;
const uint64_t FR_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
const uint64_t FR_INV = 0xc2e1f593efffffffULL;
const uint64_t FR_ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
const uint64_t FR_R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
// ark-bn254 FrConfig::TWO_ADIC_ROOT_OF_UNITY = 5^((r-1)/2^28), Montgomery form; TWO_ADICITY = 28
const uint64_t FR_TWO_ADIC_ROOT[4] = {0x636e735580d13d9cULL, 0xa22bf3742445ffd6ULL, 0x56452ac01eb203d8ULL, 0x1860ef942963f9e7ULL};
const unsigned FR_TWO_ADICITY = 28;

bool geq_mod(const uint64_t a[4]) {
  for (int i = 3; i >= 0; i--) { if (a[i] > FR_MOD[i]) return true; if (a[i] < FR_MOD[i]) return false; }
  return true;
}
void sub_mod_raw(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - b[i] - (uint64_t)br; r[i] = (uint64_t)t; br = (t >> 64) & 1; }
}
void mont_mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * FR_INV;
    c = (u128)m * FR_MOD[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * FR_MOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || geq_mod(t)) sub_mod_raw(t, t, FR_MOD);
  memcpy(r, t, 32);
}
}  // namespace

Fr Fr::one() { Fr r; memcpy(r.l, FR_ONE, 32); return r; }
Fr Fr::from_u64(uint64_t v) { Fr a; a.l[0] = v; Fr r; mont_mul(r.l, a.l, FR_R2); return r; }
Fr Fr::from_i64(int64_t v) { return v >= 0 ? from_u64((uint64_t)v) : -from_u64((uint64_t)(-(v + 1)) + 1); }
Fr Fr::operator+(const Fr& o) const {
  Fr r; u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)l[i] + o.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  if (c || geq_mod(r.l)) sub_mod_raw(r.l, r.l, FR_MOD);
  return r;
}
Fr Fr::operator-(const Fr& o) const {
  Fr r; u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)l[i] - o.l[i] - (uint64_t)br; r.l[i] = (uint64_t)t; br = (t >> 64) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + FR_MOD[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
  return r;
}
Fr Fr::operator*(const Fr& o) const { Fr r; mont_mul(r.l, l, o.l); return r; }
Fr Fr::operator-() const { return is_zero() ? *this : Fr::zero() - *this; }
Fr Fr::pow(uint64_t e) const {
  Fr acc = Fr::one(), b = *this;
  while (e) { if (e & 1) acc = acc * b; b = b * b; e >>= 1; }
  return acc;
}
Fr Fr::inverse() const {
  uint64_t e[4], two[4] = {2, 0, 0, 0};
  sub_mod_raw(e, FR_MOD, two);
  Fr acc = Fr::one();
  for (int i = 255; i >= 0; i--) { acc = acc * acc; if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * *this; }
  return acc;
}
Fr fr_rand(Rng& rng) {
  for (;;) {
    Fr r;
    for (int i = 0; i < 4; i++) r.l[i] = rng.next_u64();
    r.l[3] &= 0xFFFFFFFFFFFFFFFFULL >> 2;
    if (!geq_mod(r.l)) return r;
  }
}

// ------------------------------------------------------------------------------------------------ Device
Device::Device(int ordinal) {
  int st = keaki_hip_ctx_create(ordinal, nullptr, &ctx_);
  if (st != KEAKI_OK) throw HipError(st, keaki_hip_last_error(nullptr));
}
Device::Device(const std::vector<int>& ordinals) {
  std::vector<int32_t> d(ordinals.begin(), ordinals.end());
  int st = keaki_hip_group_create(d.data(), d.size(), &group_);
  if (st != KEAKI_OK) throw HipError(st, keaki_hip_group_last_error(nullptr));
  ctx_ = keaki_hip_group_ctx(group_, 0);       // owned by the group
}
Device::~Device() {
  if (group_) keaki_hip_group_destroy(group_);
  else keaki_hip_ctx_destroy(ctx_);
}
size_t Device::members() const { return group_ ? keaki_hip_group_size(group_) : 1; }
void Device::check(int status) const {
  if (status != KEAKI_OK) throw HipError(status, keaki_hip_last_error(ctx_));
}
void Device::check_group(int status) const {
  if (status != KEAKI_OK) throw HipError(status, keaki_hip_group_last_error(group_));
}

namespace {
// Montgomery limbs of the BN254 generators (ark-bn254 g1.rs / g2.rs): G1 = (1, 2)
const uint64_t G1_GEN_W[8] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL,
                              0xa6ba871b8b1e1b3aULL, 0x14f1d651eb8e167bULL, 0xccdd46def0f28c58ULL, 0x1c14ef83340fbe5eULL};
G1 g1_generator() { G1 g; memcpy(g.w.data(), G1_GEN_W, 64); return g; }
G2 g2_generator(const Device& dev) {
  // obtained from the device library so there is one source of truth for the G2 generator:
  // 1 * g2 via the fixed-generator path of encap (ct = r * (tau_g2 - 0 * g2) with tau_g2 = ... ) would be
  // roundabout; instead keep the constant here.
  static const uint64_t W[16] = {0x8e83b5d102bc2026ULL, 0xdceb1935497b0172ULL, 0xfbb8264797811adfULL, 0x19573841af96503bULL,
                                 0xafb4737da84c6140ULL, 0x6043dd5a5802d8c4ULL, 0x09e950fc52a02f86ULL, 0x14fef0833aea7b6bULL,
                                 0x619dfa9d886be9f6ULL, 0xfe7fd297f59e9b78ULL, 0xff9e1a62231b7dfeULL, 0x28fd7eebae9e4206ULL,
                                 0x64095b56c71856eeULL, 0xdc57f922327d3cbbULL, 0x55f935be33351076ULL, 0x0da4a0e693fd6482ULL};
  (void)dev;
  G2 g; memcpy(g.w.data(), W, 128); return g;
}
// normalised Jacobian (x, y, 1 | 1, 1, 0) -> affine encoding
G1 jac_to_g1(const uint64_t j[12]) {
  G1 r;
  if (j[8] | j[9] | j[10] | j[11]) memcpy(r.w.data(), j, 64);
  return r;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ kzg
namespace kzg {

std::string KZGError::to_string() const {
  return "Can't commit to polynomial: polynomial has degree " + std::to_string(degree) + " but max degree is " + std::to_string(max_degree);
}

KZGSetup::~KZGSetup() {
  if (chunk_ && dev_) keaki_hip_srs_g1_free(dev_->ctx(), chunk_);
  if (srs_ && dev_) keaki_hip_srs_g1_free(dev_->ctx(), srs_);
  if (gsrs_ && dev_) keaki_hip_group_srs_g1_free(dev_->group(), gsrs_);
  if (gfk_ && dev_) keaki_hip_group_fk_free(dev_->group(), gfk_);
}
KZGSetup::KZGSetup(KZGSetup&& o) noexcept
    : dev_(std::move(o.dev_)), g1_aff_(std::move(o.g1_aff_)), tau_g2_(o.tau_g2_), srs_(o.srs_), gsrs_(o.gsrs_), gfk_(o.gfk_), gfk_log2d_(o.gfk_log2d_),
      tables_(o.tables_), chunk_(o.chunk_), chunk_lo_(o.chunk_lo_), chunk_hi_(o.chunk_hi_) {
  o.srs_ = nullptr; o.chunk_ = nullptr; o.gsrs_ = nullptr; o.gfk_ = nullptr;
}
keaki_hip_group_fk* KZGSetup::group_fk(unsigned log2d) const {
  if (gfk_ && gfk_log2d_ == log2d) return gfk_;
  if (gfk_) { keaki_hip_group_fk_free(dev_->group(), gfk_); gfk_ = nullptr; }
  vec::Radix2Domain d2 = vec::Radix2Domain::create((size_t)2 << log2d);
  dev_->check_group(keaki_hip_group_fk_create(dev_->group(), g1_aff_[0].w.data(), log2d, d2.group_gen.l, d2.group_gen_inv.l, d2.size_inv.l, &gfk_));
  gfk_log2d_ = log2d;
  return gfk_;
}
keaki_hip_srs_g1* KZGSetup::srs() const {
  if (!srs_) dev_->check(keaki_hip_srs_g1_upload(dev_->ctx(), g1_aff_.empty() ? nullptr : g1_aff_[0].w.data(), g1_aff_.size(), &srs_));
  return srs_;
}

// window tables are an optimisation: when they do not fit (KEAKI_ERR_OOM) the handle keeps working through the generic path
static bool try_precompute(const Device& dev, keaki_hip_srs_g1* srs) {
  const int st = keaki_hip_srs_g1_precompute(dev.ctx(), srs, nullptr);
  if (st == KEAKI_ERR_OOM) return false;
  dev.check(st);
  return true;
}

keaki_hip_srs_g1* KZGSetup::chunk_srs(size_t lo, size_t hi) const {
  if (lo == 0 && hi == g1_aff_.size()) return srs();            // one rank: the chunk is the SRS, its tables are there already
  if (chunk_ && chunk_lo_ == lo && chunk_hi_ == hi) return chunk_;
  if (chunk_) { keaki_hip_srs_g1_free(dev_->ctx(), chunk_); chunk_ = nullptr; }
  dev_->check(keaki_hip_srs_g1_slice(dev_->ctx(), srs(), lo, hi - lo, &chunk_));
  chunk_lo_ = lo; chunk_hi_ = hi;
  if (tables_ && hi > lo) (void)try_precompute(*dev_, chunk_);
  return chunk_;
}

KZGSetup KZGSetup::from_powers(std::shared_ptr<Device> dev, std::vector<G1> g1_aff, const G2& tau_g2) {
  KZGSetup s;
  s.dev_ = std::move(dev);
  s.g1_aff_ = std::move(g1_aff);
  s.tau_g2_ = tau_g2;
  if (s.dev_->group()) {
    // several GPUs in this process: chunk i of the SRS and its window tables live on member i from now on (src/kzg.rs:98 reads them)
    s.dev_->check_group(keaki_hip_group_srs_g1_upload(s.dev_->group(), s.g1_aff_.empty() ? nullptr : s.g1_aff_[0].w.data(), s.g1_aff_.size(), 1, &s.gsrs_));
    s.tables_ = keaki_hip_group_srs_g1_has_tables(s.gsrs_) != 0;    // a member whose table build ran out of memory keeps the generic MSM
    return s;
  }
  s.dev_->check(keaki_hip_srs_g1_upload(s.dev_->ctx(), s.g1_aff_.empty() ? nullptr : s.g1_aff_[0].w.data(), s.g1_aff_.size(), &s.srs_));
  // The SRS never changes after setup: tabulate [2^(window offset)] tau^i G1 once so that every later commit/open runs the
  // shared-bucket MSM (W x the SRS in HBM). It pays at every size: a 1000-coefficient commit takes 0.64 ms with tables, 1.98 ms
  // without (the per-window sums need ~250 serial doublings). KEAKI_PRECOMPUTE_MIN raises the smallest SRS that gets tables.
  const char* pm = getenv("KEAKI_PRECOMPUTE_MIN");
  const size_t pre_min = pm ? (size_t)atoll(pm) : (size_t)1;
  if (s.g1_aff_.size() >= pre_min) s.tables_ = try_precompute(*s.dev_, s.srs_);
  return s;
}

KZGSetup KZGSetup::setup(std::shared_ptr<Device> dev, const Fr& secret, size_t max_d) {
  // g1_pow[i] = g1 * secret^i (src/kzg.rs:59-61), tau_g2 = g2 * secret (:57); batched on the GPU
  std::vector<Fr> pw(max_d);
  Fr acc = Fr::one();
  for (size_t i = 0; i < max_d; i++) { pw[i] = acc; acc = acc * secret; }
  std::vector<G1> pts(max_d);
  G1 g1 = g1_generator();
  if (max_d) dev->check(keaki_hip_g1_mul_batch(dev->ctx(), g1.w.data(), 0, pw[0].l, max_d, pts[0].w.data()));
  G2 g2 = g2_generator(*dev), tau;
  dev->check(keaki_hip_g2_mul_batch(dev->ctx(), g2.w.data(), 0, secret.l, 1, tau.w.data()));
  return from_powers(std::move(dev), std::move(pts), tau);
}

Result<G1> commit(const KZGSetup& setup, const DensePolynomial& p) {
  if (p.size() > setup.g1_pow().size())
    return Result<G1>::Err(KZGError{KZGError::PolynomialTooLarge, p.size(), setup.g1_pow().size()});
  uint64_t jac[12];
  if (setup.group_srs())
    setup.device()->check_group(keaki_hip_group_msm_g1(setup.device()->group(), setup.group_srs(), p.empty() ? nullptr : p[0].l, p.size(), jac));
  else
    setup.device()->check(keaki_hip_msm_g1(setup.device()->ctx(), setup.srs(), p.empty() ? nullptr : p[0].l, p.size(), jac));
  return Result<G1>::Ok(jac_to_g1(jac));
}

constexpr size_t OPEN_HOST_QUOTIENT_MAX = 8192;
// DensePolynomial semantics: trailing zero coefficients are not part of the polynomial
static void trim(DensePolynomial& p) { while (!p.empty() && p.back().is_zero()) p.pop_back(); }

Result<G1> open(const KZGSetup& setup, const DensePolynomial& p_in, const Fr& point) {
  // quotient (p(x) - p(point)) / (x - point) and its commitment in one device call (keaki_hip_kzg_open: blockwise Horner recurrence
  // in front of the MSM). The quotient has p.len() - 1 coefficients and its leading one is p's, so it needs no trimming.
  DensePolynomial p = p_in;
  trim(p);
  const size_t qlen = p.size() > 1 ? p.size() - 1 : 0;
  if (qlen > setup.g1_pow().size())                       // the commit inside open fails with the QUOTIENT's length (src/kzg.rs:123,91-96)
    return Result<G1>::Err(KZGError{KZGError::PolynomialTooLarge, qlen, setup.g1_pow().size()});
  uint64_t jac[12];
  if (setup.group_srs()) {
    setup.device()->check_group(keaki_hip_group_kzg_open(setup.device()->group(), setup.group_srs(), p.empty() ? nullptr : p[0].l, p.size(), point.l, jac, nullptr));
  } else if (qlen && qlen <= OPEN_HOST_QUOTIENT_MAX) {
    // SHORT polynomials: the synthetic division is a chain (q_(i-1) = p_i + point q_i), 40 ns a step on the host, while the device's blockwise
    // recurrence waits for its three launches (0.7 ms at 1,000 coefficients against 0.04 here); the MSM is the device's either way (src/kzg.rs:109-123)
    std::vector<Fr> q(qlen);
    q[qlen - 1] = p[qlen];
    for (size_t i = qlen - 1; i > 0; i--) q[i - 1] = p[i] + point * q[i];
    setup.device()->check(keaki_hip_msm_g1(setup.device()->ctx(), setup.srs(), q[0].l, qlen, jac));
  } else {
    setup.device()->check(keaki_hip_kzg_open(setup.device()->ctx(), setup.srs(), p.empty() ? nullptr : p[0].l, p.size(), point.l, jac, nullptr));
  }
  return Result<G1>::Ok(jac_to_g1(jac));
}

Result<bool> verify(const KZGSetup& setup, const G1& commitment, const Fr& point, const Fr& value, const G1& proof) {
  // src/kzg.rs:127-146, evaluated on the device in the same form (include/keaki_hip.h)
  const Device& dev = *setup.device();
  int32_t ok = 0;
  dev.check(keaki_hip_kzg_verify(dev.ctx(), commitment.w.data(), setup.tau_g2().w.data(), point.l, value.l, proof.w.data(), &ok));
  return Result<bool>::Ok(ok != 0);
}

void precompute_open_fk(const KZGSetup& setup, size_t d) {
  if (d < 1 || (d & (d - 1)) != 0 || d > setup.g1_pow().size()) return;   // open_fk falls back to per-point openings for such shapes
  unsigned log2d = 0;
  while ((size_t(1) << log2d) < d) log2d++;
  if (setup.device()->group()) { (void)setup.group_fk(log2d); return; }      // every member's part of the SRS-only transform
  vec::Radix2Domain d2 = vec::Radix2Domain::create(2 * d);
  setup.device()->check(keaki_hip_srs_g1_precompute_fk(setup.device()->ctx(), setup.srs(), log2d, d2.group_gen.l));
}

Result<std::vector<G1>> open_fk(const KZGSetup& setup, const std::vector<Fr>& p, size_t domain_size) {
  const size_t d = p.size();
  const bool pow2 = d >= 1 && (d & (d - 1)) == 0;
  if (pow2 && domain_size == d && d <= setup.g1_pow().size()) {
    // FK23 (src/kzg.rs:157-203): the scalar-field DFT, the three group FFTs and the 2d scalar-mults all run on the GPU.
    unsigned log2d = 0;
    while ((size_t(1) << log2d) < d) log2d++;
    // hat_a = DFT_2d(0,..,0,p) / 2d and every twiddle table are derived on the device from omega_2d (keaki_hip_open_fk_poly)
    vec::Radix2Domain d2 = vec::Radix2Domain::create(2 * d);
    std::vector<G1> out(d);
    const Device& dev = *setup.device();
    if (dev.group())       // the group FFTs and the 2d scalar-mults split over the members, exchanges inside the library
      dev.check_group(keaki_hip_group_fk_open(dev.group(), setup.group_fk(log2d), p[0].l, out[0].w.data()));
    else
      dev.check(keaki_hip_open_fk_poly(dev.ctx(), setup.srs(), log2d, p[0].l, d2.group_gen.l, d2.group_gen_inv.l, d2.size_inv.l, out[0].w.data()));
    return Result<std::vector<G1>>::Ok(std::move(out));
  }
  // shapes FK23 does not cover (the reference would panic on them): one opening per root of unity
  vec::Radix2Domain dd = vec::Radix2Domain::create(domain_size);
  std::vector<Fr> el = dd.elements();
  std::vector<G1> out;
  out.reserve(el.size());
  for (const Fr& z : el) {
    Result<G1> r = open(setup, p, z);
    if (!r.ok) return Result<std::vector<G1>>::Err(r.error);
    out.push_back(r.value);
  }
  return Result<std::vector<G1>>::Ok(std::move(out));
}

}  // namespace kzg

// ------------------------------------------------------------------------------------------------ kem / enc
namespace kem {

std::pair<G2, std::vector<uint8_t>> encapsulate(Rng& rng, const kzg::KZGSetup& setup, const G1& commitment, const Fr& point,
                                                const Fr& value, size_t msg_len) {
  Fr r = fr_rand(rng);  // src/kem.rs:26
  G2 ct; std::vector<uint8_t> key(msg_len), gt(384);
  setup.device()->check(keaki_hip_encap_batch(setup.device()->ctx(), commitment.w.data(), setup.tau_g2().w.data(), point.l, value.l, r.l, 1,
                                              ct.w.data(), gt.data(), msg_len ? key.data() : nullptr, msg_len));
  return {ct, key};
}
std::vector<uint8_t> decapsulate(const kzg::KZGSetup& setup, const G1& proof, const G2& ciphertext, size_t msg_len) {
  std::vector<uint8_t> key(msg_len), gt(384);
  setup.device()->check(keaki_hip_decap_batch(setup.device()->ctx(), proof.w.data(), ciphertext.w.data(), 1, gt.data(),
                                              msg_len ? key.data() : nullptr, msg_len));
  return key;
}

void prepare(const kzg::KZGSetup& setup, size_t batch_hint) {
  const Device& dev = *setup.device();
  if (dev.group() && batch_hint >= GROUP_MIN_ITEMS) {
    const size_t N = dev.members(), share = (batch_hint + N - 1) / N;
    for (size_t i = 0; i < N; i++) {
      keaki_hip_ctx* c = keaki_hip_group_ctx(dev.group(), i);
      const int st = keaki_hip_encap_prepare(c, setup.tau_g2().w.data(), share);
      if (st != KEAKI_OK) throw HipError(st, keaki_hip_last_error(c));
    }
    return;
  }
  dev.check(keaki_hip_encap_prepare(dev.ctx(), setup.tau_g2().w.data(), batch_hint));
}
}  // namespace kem

namespace enc {
Ciphertext encrypt(Rng& rng, const kzg::KZGSetup& setup, const G1& com, const Fr& point, const Fr& value, const std::vector<uint8_t>& msg) {
  auto kc = kem::encapsulate(rng, setup, com, point, value, msg.size());
  std::vector<uint8_t> ct(msg.size());
  for (size_t i = 0; i < msg.size(); i++) ct[i] = kc.second[i] ^ msg[i];
  return {kc.first, ct};
}
std::vector<uint8_t> decrypt(const kzg::KZGSetup& setup, const G1& proof, const Ciphertext& ct) {
  std::vector<uint8_t> key = kem::decapsulate(setup, proof, ct.first, ct.second.size());
  for (size_t i = 0; i < key.size(); i++) key[i] ^= ct.second[i];
  return key;
}
}  // namespace enc

// ------------------------------------------------------------------------------------------------ vec
namespace vec {
namespace {
// the loop bodies of src/vec.rs:63-66 / :75-78 as one batched call: over all members of a group device for large batches
void encap_many(const kzg::KZGSetup& setup, const G1& com, const uint64_t* points, const uint64_t* values, const uint64_t* rs, size_t n,
                uint64_t* ct, uint8_t* key, size_t msg_len) {
  const Device& dev = *setup.device();
  if (dev.group() && n >= GROUP_MIN_ITEMS)
    dev.check_group(keaki_hip_group_encap_batch(dev.group(), com.w.data(), setup.tau_g2().w.data(), points, values, rs, n, ct, nullptr, key, msg_len));
  else
    dev.check(keaki_hip_encap_batch(dev.ctx(), com.w.data(), setup.tau_g2().w.data(), points, values, rs, n, ct, nullptr, key, msg_len));
}
// enc::encrypt / enc::decrypt per item (src/enc.rs:19-55) as one batched call: KEM + the XOR DEM on the device(s)
void encrypt_many(const kzg::KZGSetup& setup, const G1& com, const uint64_t* points, const uint64_t* values, const uint64_t* rs, const uint8_t* msgs,
                  size_t n, uint64_t* ct, uint8_t* body, size_t msg_len) {
  const Device& dev = *setup.device();
  if (dev.group() && n >= GROUP_MIN_ITEMS)
    dev.check_group(keaki_hip_group_encrypt_batch(dev.group(), com.w.data(), setup.tau_g2().w.data(), points, values, rs, msgs, n, ct, body, msg_len));
  else
    dev.check(keaki_hip_encrypt_batch(dev.ctx(), com.w.data(), setup.tau_g2().w.data(), points, values, rs, msgs, n, ct, body, msg_len));
}
void decrypt_many(const kzg::KZGSetup& setup, const uint64_t* proofs, const uint64_t* cts, const uint8_t* bodies, size_t n, uint8_t* msgs, size_t msg_len) {
  const Device& dev = *setup.device();
  if (dev.group() && n >= GROUP_MIN_ITEMS) dev.check_group(keaki_hip_group_decrypt_batch(dev.group(), proofs, cts, bodies, n, msgs, msg_len));
  else dev.check(keaki_hip_decrypt_batch(dev.ctx(), proofs, cts, bodies, n, msgs, msg_len));
}
void decap_many(const kzg::KZGSetup& setup, const uint64_t* proofs, const uint64_t* cts, size_t n, uint8_t* key, size_t msg_len) {
  const Device& dev = *setup.device();
  if (dev.group() && n >= GROUP_MIN_ITEMS)
    dev.check_group(keaki_hip_group_decap_batch(dev.group(), proofs, cts, n, nullptr, key, msg_len));   // keys only: no 384 B/item GT download
  else
    dev.check(keaki_hip_decap_batch(dev.ctx(), proofs, cts, n, nullptr, key, msg_len));
}
}  // namespace

Radix2Domain Radix2Domain::create(size_t min_size) {
  size_t size = 1; unsigned log = 0;
  while (size < min_size) { size <<= 1; log++; }
  if (log > FR_TWO_ADICITY) throw std::invalid_argument("domain larger than 2^28");
  Fr g; memcpy(g.l, FR_TWO_ADIC_ROOT, 32);
  for (unsigned i = log; i < FR_TWO_ADICITY; i++) g = g * g;
  Radix2Domain d;
  d.size = size; d.group_gen = g; d.group_gen_inv = g.inverse(); d.size_inv = Fr::from_u64(size).inverse();
  return d;
}
std::vector<Fr> Radix2Domain::elements() const {
  std::vector<Fr> e(size);
  Fr acc = Fr::one();
  for (size_t i = 0; i < size; i++) { e[i] = acc; acc = acc * group_gen; }
  return e;
}
static void fft_in_place(std::vector<Fr>& a, const Fr& root) {
  size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; i++) {  // bit reversal
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    Fr w = root;
    for (size_t k = len; k < n; k <<= 1) w = w * w;
    for (size_t i = 0; i < n; i += len) {
      Fr x = Fr::one();
      for (size_t j = 0; j < len / 2; j++) {
        Fr u = a[i + j], v = a[i + j + len / 2] * x;
        a[i + j] = u + v; a[i + j + len / 2] = u - v;
        x = x * w;
      }
    }
  }
}
std::vector<Fr> Radix2Domain::fft(std::vector<Fr> c) const { c.resize(size); fft_in_place(c, group_gen); return c; }
std::vector<Fr> Radix2Domain::ifft(std::vector<Fr> e) const {
  e.resize(size);
  fft_in_place(e, group_gen_inv);
  for (auto& x : e) x = x * size_inv;
  return e;
}

// src/vec.rs:27-44: everything of vec_commit in front of the final commit -> (coefficients as a DensePolynomial, proofs)
std::vector<Fr> vec_commit_coeffs(Rng& rng, const kzg::KZGSetup& setup, const std::vector<Fr>& v) {
  size_t d = v.size() + PADDING_LEN;
  std::vector<Fr> padded(v);
  padded.push_back(fr_rand(rng));                    // src/vec.rs:31-33
  Radix2Domain domain = Radix2Domain::create(d);     // :36
  if (d < (size_t(1) << 12)) return domain.ifft(padded);
  // :37 on the device for large domains (keaki_hip_fr_fft), same values
  padded.resize(domain.size);
  unsigned log_size = 0;
  while ((size_t(1) << log_size) < domain.size) log_size++;
  setup.device()->check(keaki_hip_fr_fft(setup.device()->ctx(), padded[0].l, log_size, domain.group_gen_inv.l, domain.size_inv.l));
  return padded;
}
static DensePolynomial trimmed(std::vector<Fr> c) {
  while (!c.empty() && c.back().is_zero()) c.pop_back();  // from_coefficients_vec trims
  return c;
}
std::pair<DensePolynomial, std::vector<G1>> vec_commit_openings(Rng& rng, const kzg::KZGSetup& setup, const std::vector<Fr>& v) {
  std::vector<Fr> p_coeff = vec_commit_coeffs(rng, setup, v);
  std::vector<G1> proofs = kzg::open_fk(setup, p_coeff, p_coeff.size()).unwrap();  // :40
  return {trimmed(std::move(p_coeff)), std::move(proofs)};
}
G1 vec_commit_flat(Rng& rng, const kzg::KZGSetup& setup, const Fr* v, size_t n, uint64_t* proofs_out) {
  const size_t d = n + PADDING_LEN;
  Radix2Domain domain = Radix2Domain::create(d);     // src/vec.rs:36
  const Device& dev = *setup.device();
  if (!dev.group() && domain.size >= (size_t(1) << 12) && domain.size <= setup.g1_pow().size()) {
    const Fr pad = fr_rand(rng);                     // :31-33, the only draw
    unsigned log_size = 0;
    while ((size_t(1) << log_size) < domain.size) log_size++;
    Radix2Domain d2 = Radix2Domain::create(2 * domain.size);
    uint64_t jac[12];
    dev.check(keaki_hip_vec_commit(dev.ctx(), setup.srs(), n ? v[0].l : nullptr, n, pad.l, log_size, domain.group_gen_inv.l, domain.size_inv.l,
                                   d2.group_gen.l, d2.group_gen_inv.l, d2.size_inv.l, jac, proofs_out));
    return jac_to_g1(jac);
  }
  auto cp = vec_commit_openings(rng, setup, std::vector<Fr>(v, v + n));
  for (size_t i = 0; i < cp.second.size(); i++) memcpy(proofs_out + 8 * i, cp.second[i].w.data(), 64);
  return kzg::commit(setup, cp.first).unwrap();      // :46
}
std::pair<G1, std::vector<G1>> vec_commit(Rng& rng, const kzg::KZGSetup& setup, const std::vector<Fr>& v) {
  static_assert(sizeof(G1) == 64, "a G1 is eight u64 words");
  std::vector<G1> proofs(Radix2Domain::create(v.size() + PADDING_LEN).size);
  G1 com = vec_commit_flat(rng, setup, v.data(), v.size(), proofs.empty() ? nullptr : proofs[0].w.data());
  return {com, std::move(proofs)};
}

std::vector<enc::Ciphertext> vec_encrypt(Rng& rng, const kzg::KZGSetup& setup, const G1& com, const std::vector<Fr>& points,
                                         const std::vector<Fr>& values, const std::vector<std::vector<uint8_t>>& messages) {
  size_t n = messages.size();
  std::vector<enc::Ciphertext> out(n);
  if (!n) return out;
  // the reference draws one r per item in index order inside the loop (src/vec.rs:63-66 -> src/kem.rs:26)
  std::vector<Fr> rs(n);
  for (size_t i = 0; i < n; i++) rs[i] = fr_rand(rng);
  size_t max_len = 0;
  for (auto& m : messages) max_len = std::max(max_len, m.size());
  std::vector<uint64_t> ct(16 * n);
  std::vector<uint8_t> key(n * std::max<size_t>(max_len, 1));
  encap_many(setup, com, points[0].l, values[0].l, rs[0].l, n, ct.data(), key.data(), max_len);
  for (size_t i = 0; i < n; i++) {
    memcpy(out[i].first.w.data(), &ct[16 * i], 128);
    out[i].second.resize(messages[i].size());
    // BLAKE3 XOF: a shorter key is a prefix of a longer one, so one max_len call serves ragged messages
    for (size_t j = 0; j < messages[i].size(); j++) out[i].second[j] = key[i * max_len + j] ^ messages[i][j];
  }
  return out;
}

// The same two loops on contiguous buffers of n equal-length messages: identical draws and bytes, no per-item containers (what an FFI
// caller with 2^20 items wants).
void vec_encrypt_flat(Rng& rng, const kzg::KZGSetup& setup, const G1& com, const Fr* points, const Fr* values, const uint8_t* msgs, size_t n,
                      size_t msg_len, uint64_t* ct_g2_out, uint8_t* ct_msg_out) {
  if (!n) return;
  // One r per item, in index order (src/vec.rs:63-66 -> src/kem.rs:26): the draws are the host's and serial by the reference's semantics
  // (13.5 ms for 2^20 items, twice the rate of the device). Large vectors go in PIECES of growing size -- 1/16, 1/8, 1/4, the rest: a helper
  // thread draws the r of piece k + 1 while the device works on piece k, and only the first, small piece is drawn with the device idle
  // (the stream of draws is the same; only one thread touches `rng` at a time). Inside a piece the C ABI overlaps its own copies with its
  // kernels (api.hip: pipelined).
  static const size_t SIXTEENTHS[5] = {0, 1, 3, 7, 16};
  const size_t pieces = n >= ((size_t)1 << 18) ? 4 : 1;
  auto bound = [&](size_t k) { return k >= pieces ? n : std::min(n, ((n * SIXTEENTHS[k] / 16) + 63) & ~(size_t)63); };
  // the draws land in memory nobody has touched or cleared: a value-initialised std::vector<Fr>(n) spends 6 ms at n = 2^20 writing zeros into
  // fresh 4 KB pages before the first draw. Huge pages where the kernel offers them on request (the first touch is the helper thread's).
  struct RawFr {
    Fr* p = nullptr;
    explicit RawFr(size_t count) {
      const size_t HP = (size_t)2 << 20, bytes = count * sizeof(Fr);
      if (bytes >= HP) {
        p = static_cast<Fr*>(aligned_alloc(HP, (bytes + HP - 1) & ~(HP - 1)));
        if (p) (void)madvise(p, (bytes + HP - 1) & ~(HP - 1), MADV_HUGEPAGE);
      } else {
        p = static_cast<Fr*>(malloc(bytes));
      }
      if (!p) throw std::bad_alloc();
    }
    ~RawFr() { free(p); }
    RawFr(const RawFr&) = delete;
    RawFr& operator=(const RawFr&) = delete;
  } rs_mem(n);
  Fr* rs = rs_mem.p;
  auto draw = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) rs[i] = fr_rand(rng); };
  std::vector<uint8_t> gt_unused;
  draw(0, bound(1));
  for (size_t k = 0; k < pieces; k++) {
    const size_t lo = bound(k), hi = bound(k + 1), m = hi - lo;
    std::thread ahead;
    if (k + 1 < pieces) ahead = std::thread(draw, hi, bound(k + 2));
    try {
      if (m && msg_len)                                                                                         // src/enc.rs:32-36 on the device
        encrypt_many(setup, com, points[lo].l, values[lo].l, rs[lo].l, msgs + msg_len * lo, m, ct_g2_out + 16 * lo, ct_msg_out + msg_len * lo, msg_len);
      else if (m) {
        gt_unused.resize(m * 384);       // the ABI wants at least one of gt / key
        setup.device()->check(keaki_hip_encap_batch(setup.device()->ctx(), com.w.data(), setup.tau_g2().w.data(), points[lo].l, values[lo].l, rs[lo].l, m,
                                                    ct_g2_out + 16 * lo, gt_unused.data(), nullptr, 0));
      }
    } catch (...) {
      if (ahead.joinable()) ahead.join();
      throw;
    }
    if (ahead.joinable()) ahead.join();
  }
}
void vec_decrypt_flat(const kzg::KZGSetup& setup, const uint64_t* proofs, const uint64_t* ct_g2, const uint8_t* ct_msgs, size_t n, size_t msg_len,
                      uint8_t* msgs_out) {
  if (!n || !msg_len) return;
  decrypt_many(setup, proofs, ct_g2, ct_msgs, n, msgs_out, msg_len);                                          // src/enc.rs:48-52 on the device
}
std::vector<std::vector<uint8_t>> vec_decrypt(const kzg::KZGSetup& setup, const std::vector<G1>& proofs,
                                              const std::vector<const enc::Ciphertext*>& cts) {
  size_t n = cts.size();
  std::vector<std::vector<uint8_t>> out(n);
  if (!n) return out;
  size_t max_len = 0;
  for (auto* c : cts) max_len = std::max(max_len, c->second.size());
  std::vector<uint64_t> pr(8 * n), ct(16 * n);
  for (size_t i = 0; i < n; i++) { memcpy(&pr[8 * i], proofs[i].w.data(), 64); memcpy(&ct[16 * i], cts[i]->first.w.data(), 128); }
  std::vector<uint8_t> key(n * std::max<size_t>(max_len, 1));
  decap_many(setup, pr.data(), ct.data(), n, key.data(), max_len);
  for (size_t i = 0; i < n; i++) {
    out[i].resize(cts[i]->second.size());
    for (size_t j = 0; j < out[i].size(); j++) out[i][j] = key[i * max_len + j] ^ cts[i]->second[j];
  }
  return out;
}

}  // namespace vec

// ------------------------------------------------------------------------------------------------ dist
namespace dist {

std::pair<size_t, size_t> Shard::bounds(size_t n) const {
  const size_t base = n / world, rem = n % world;
  const size_t lo = rank * base + std::min(rank, rem);
  return {lo, lo + base + (rank < rem ? 1 : 0)};
}

void prepare(const kzg::KZGSetup& setup, const Shard& sh) {
  auto b = sh.bounds(setup.g1_pow().size());
  (void)setup.chunk_srs(b.first, b.second);
}

kzg::Result<Partial> commit_partial(const kzg::KZGSetup& setup, const DensePolynomial& p, const Shard& sh) {
  const size_t len = setup.g1_pow().size();
  if (p.size() > len) return kzg::Result<Partial>::Err(kzg::KZGError{kzg::KZGError::PolynomialTooLarge, p.size(), len});
  // the ranges follow the SRS, not the polynomial: a rank's window tables are built once per setup and serve every polynomial
  auto b = sh.bounds(len);
  const size_t lo = std::min(b.first, p.size()), hi = std::min(b.second, p.size());
  Partial out;
  setup.device()->check(keaki_hip_msm_g1(setup.device()->ctx(), setup.chunk_srs(b.first, b.second), hi > lo ? p[lo].l : nullptr, hi - lo, out.data()));
  return kzg::Result<Partial>::Ok(out);
}

std::pair<Partial, std::vector<G1>> vec_commit_partial(Rng& rng, const kzg::KZGSetup& setup, const std::vector<Fr>& v, const Shard& sh) {
  auto cp = vec::vec_commit_openings(rng, setup, v);                 // replicated on every rank: same rng stream, same values
  return {commit_partial(setup, cp.first, sh).unwrap(), std::move(cp.second)};
}

static unsigned log2_of(size_t n) {
  unsigned l = 0;
  while ((size_t(1) << l) < n) l++;
  return l;
}
bool ShardedOpenFk::can_shard(const kzg::KZGSetup& setup, size_t domain_size, const Shard& sh) {
  const size_t w = sh.world;
  return w >= 2 && (w & (w - 1)) == 0 && domain_size >= w * w && (domain_size & (domain_size - 1)) == 0 && domain_size <= setup.g1_pow().size();
}
ShardedOpenFk::ShardedOpenFk(const kzg::KZGSetup& setup, size_t domain_size, const Shard& sh) : setup_(setup), dev_(setup.device()), d_(domain_size) {
  vec::Radix2Domain d2 = vec::Radix2Domain::create(2 * domain_size);
  const Device& dev = *setup.device();
  dev.check(keaki_hip_fk_shard_create(dev.ctx(), setup.srs(), log2_of(domain_size), (uint32_t)sh.rank, (uint32_t)sh.world, d2.group_gen.l,
                                      d2.group_gen_inv.l, d2.size_inv.l, &fk_));
  dev.check(keaki_hip_fk_shard_sizes(fk_, sizes_));
}
ShardedOpenFk::~ShardedOpenFk() { keaki_hip_fk_shard_free(dev_->ctx(), fk_); }
void ShardedOpenFk::prepare(void* d_send, void* d_recv, const FkExchange& ex) {
  if (prepared_) return;
  const Device& dev = *setup_.device();
  dev.check(keaki_hip_fk_shard_setup(dev.ctx(), fk_, 0, d_send, nullptr));
  dev.check(keaki_hip_synchronize(dev.ctx()));
  if (ex.all_to_all(ex.user, d_send, d_recv, sizes_[1])) throw std::runtime_error("ShardedOpenFk: the caller's all-to-all failed");
  dev.check(keaki_hip_fk_shard_setup(dev.ctx(), fk_, 1, nullptr, d_recv));
  dev.check(keaki_hip_synchronize(dev.ctx()));
  prepared_ = true;
}
std::vector<G1> ShardedOpenFk::open(const std::vector<Fr>& p, void* d_send, void* d_recv, const FkExchange& ex) {
  if (p.size() != d_) throw std::runtime_error("ShardedOpenFk::open: the polynomial must have domain_size coefficients");
  prepare(d_send, d_recv, ex);
  const Device& dev = *setup_.device();
  dev.check(keaki_hip_fk_shard_open(dev.ctx(), fk_, 0, p[0].l, d_send, nullptr, nullptr));
  dev.check(keaki_hip_synchronize(dev.ctx()));
  if (ex.all_to_all(ex.user, d_send, d_recv, sizes_[2])) throw std::runtime_error("ShardedOpenFk: the caller's all-to-all failed");
  dev.check(keaki_hip_fk_shard_open(dev.ctx(), fk_, 1, nullptr, d_send, d_recv, nullptr));
  dev.check(keaki_hip_synchronize(dev.ctx()));
  if (ex.all_to_all(ex.user, d_send, d_recv, sizes_[2])) throw std::runtime_error("ShardedOpenFk: the caller's all-to-all failed");
  dev.check(keaki_hip_fk_shard_open(dev.ctx(), fk_, 2, nullptr, d_send, d_recv, nullptr));
  dev.check(keaki_hip_synchronize(dev.ctx()));
  if (ex.all_gather(ex.user, d_send, d_recv, sizes_[3])) throw std::runtime_error("ShardedOpenFk: the caller's all-gather failed");
  std::vector<G1> out(d_);
  dev.check(keaki_hip_fk_shard_open(dev.ctx(), fk_, 3, nullptr, nullptr, d_recv, out[0].w.data()));
  return out;
}
std::pair<Partial, std::vector<G1>> vec_commit_partial_fk(Rng& rng, const kzg::KZGSetup& setup, const std::vector<Fr>& v, const Shard& sh,
                                                          ShardedOpenFk& fk, void* d_send, void* d_recv, const FkExchange& ex) {
  std::vector<Fr> coeffs = vec::vec_commit_coeffs(rng, setup, v);       // replicated: same rng stream, same values
  std::vector<G1> proofs = fk.open(coeffs, d_send, d_recv, ex);
  while (!coeffs.empty() && coeffs.back().is_zero()) coeffs.pop_back();
  return {commit_partial(setup, coeffs, sh).unwrap(), std::move(proofs)};
}

G1 commit_combine(const kzg::KZGSetup& setup, const Partial* partials, size_t world) {
  uint64_t jac[12];
  setup.device()->check(keaki_hip_g1_sum(setup.device()->ctx(), world ? partials[0].data() : nullptr, world, jac));
  return jac_to_g1(jac);
}

void vec_encrypt_flat_shard(Rng& rng, const kzg::KZGSetup& setup, const G1& com, const Fr* points, const Fr* values, const uint8_t* msgs, size_t n,
                            size_t msg_len, const Shard& sh, uint64_t* ct_g2_out, uint8_t* ct_msg_out) {
  if (!n) return;
  std::vector<Fr> rs(n);
  for (size_t i = 0; i < n; i++) rs[i] = fr_rand(rng);        // the whole stream, as the serial loop would draw it (src/vec.rs:63-66 -> src/kem.rs:26)
  auto b = sh.bounds(n);
  const size_t lo = b.first, m = b.second - b.first;
  if (!m) return;
  setup.device()->check(keaki_hip_encap_batch(setup.device()->ctx(), com.w.data(), setup.tau_g2().w.data(), points[lo].l, values[lo].l, rs[lo].l, m,
                                              ct_g2_out, nullptr, msg_len ? ct_msg_out : nullptr, msg_len));
  for (size_t i = 0; i < m * msg_len; i++) ct_msg_out[i] ^= msgs[lo * msg_len + i];
}

}  // namespace dist
}  // namespace keaki
