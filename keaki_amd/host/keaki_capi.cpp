// Flat C front-end of the C++ host mirror (keaki.hpp) so the pytest harness and other FFI users can
// drive kzg / kem / enc / vec exactly as the reference's tests do. Errors: 0 ok, 1 = KZGError
// (degree/max written to err_out[2]), negative = HipError status; message via keaki_host_last_error().
#include <algorithm>
#include <cstring>
#include <string>

#include "keaki.hpp"
#include "ptau.hpp"

using namespace keaki;

namespace {
thread_local std::string g_err;
struct CallbackRng : Rng {
  uint64_t (*fn)(void*); void* user;
  uint64_t next_u64() override { return fn(user); }
};
struct SplitMix64Rng : Rng {  // deterministic stand-in for ark_std::test_rng() in the harness
  uint64_t s;
  explicit SplitMix64Rng(uint64_t seed) : s(seed) {}
  uint64_t next_u64() override {
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
};
struct Setup { std::shared_ptr<Device> dev; kzg::KZGSetup s; };

template <class F>
int guard(F&& f) {
  try { return f(); }
  catch (const HipError& e) { g_err = e.what(); return e.status; }
  catch (const std::exception& e) { g_err = e.what(); return -1000; }
}
Fr fr_of(const uint64_t* p) { Fr r; memcpy(r.l, p, 32); return r; }
std::vector<Fr> frs_of(const uint64_t* p, size_t n) { std::vector<Fr> v(n); for (size_t i = 0; i < n; i++) v[i] = fr_of(p + 4 * i); return v; }
G1 g1_of(const uint64_t* p) { G1 r; memcpy(r.w.data(), p, 64); return r; }
G2 g2_of(const uint64_t* p) { G2 r; memcpy(r.w.data(), p, 128); return r; }
int kzg_err(const kzg::KZGError& e, uint64_t* err_out) { if (err_out) { err_out[0] = e.degree; err_out[1] = e.max_degree; } g_err = e.to_string(); return 1; }
}  // namespace

extern "C" {

const char* keaki_host_last_error(void) { return g_err.c_str(); }

void* keaki_host_rng_splitmix(uint64_t seed) { return new SplitMix64Rng(seed); }
void* keaki_host_rng_callback(uint64_t (*fn)(void*), void* user) { auto* r = new CallbackRng(); r->fn = fn; r->user = user; return r; }
void keaki_host_rng_free(void* rng) { delete (Rng*)rng; }
void keaki_host_fr_rand(void* rng, uint64_t* out) { Fr r = fr_rand(*(Rng*)rng); memcpy(out, r.l, 32); }
// n consecutive draws (what a loop of `Fr::rand(rng)` consumes), for harnesses that replay the stream of vec_encrypt
void keaki_host_fr_rand_many(void* rng, size_t n, uint64_t* out) { for (size_t i = 0; i < n; i++) { Fr r = fr_rand(*(Rng*)rng); memcpy(out + 4 * i, r.l, 32); } }

// Fr helpers for the harness (Fr::from(i64), arithmetic, polynomial evaluation)
void keaki_host_fr_from_i64(int64_t v, uint64_t* out) { Fr r = Fr::from_i64(v); memcpy(out, r.l, 32); }
void keaki_host_fr_mul(const uint64_t* a, const uint64_t* b, uint64_t* out) { Fr r = fr_of(a) * fr_of(b); memcpy(out, r.l, 32); }
void keaki_host_fr_add(const uint64_t* a, const uint64_t* b, uint64_t* out) { Fr r = fr_of(a) + fr_of(b); memcpy(out, r.l, 32); }
void keaki_host_fr_sub(const uint64_t* a, const uint64_t* b, uint64_t* out) { Fr r = fr_of(a) - fr_of(b); memcpy(out, r.l, 32); }
void keaki_host_fr_inv(const uint64_t* a, uint64_t* out) { Fr r = fr_of(a).inverse(); memcpy(out, r.l, 32); }
void keaki_host_poly_eval(const uint64_t* coeffs, size_t n, const uint64_t* x, uint64_t* out) {
  Fr acc = Fr::zero(), xx = fr_of(x);
  for (size_t i = n; i-- > 0;) acc = acc * xx + fr_of(coeffs + 4 * i);
  memcpy(out, acc.l, 32);
}
size_t keaki_host_domain(size_t min_size, uint64_t* elements_out /* may be NULL */) {
  auto d = vec::Radix2Domain::create(min_size);
  if (elements_out) { auto e = d.elements(); for (size_t i = 0; i < e.size(); i++) memcpy(elements_out + 4 * i, e[i].l, 32); }
  return d.size;
}
void keaki_host_ifft(const uint64_t* evals, size_t n, size_t domain_min, uint64_t* out) {
  auto d = vec::Radix2Domain::create(domain_min);
  auto c = d.ifft(frs_of(evals, n));
  for (size_t i = 0; i < c.size(); i++) memcpy(out + 4 * i, c[i].l, 32);
}
void keaki_host_fft(const uint64_t* coeffs, size_t n, size_t domain_min, uint64_t* out) {
  auto d = vec::Radix2Domain::create(domain_min);
  auto c = d.fft(frs_of(coeffs, n));
  for (size_t i = 0; i < c.size(); i++) memcpy(out + 4 * i, c[i].l, 32);
}

// ---- Device objects: one GPU, or a list of ordinals = several GPUs of this process behind one object (keaki::Device) ---------------
// A handle is a std::shared_ptr<Device>*: setups created on it keep the device alive after the handle is freed.
int keaki_host_device_new(const int* ordinals, size_t n, void** out) {
  return guard([&] {
    if (n == 1) *out = new std::shared_ptr<Device>(std::make_shared<Device>(ordinals[0]));
    else *out = new std::shared_ptr<Device>(std::make_shared<Device>(std::vector<int>(ordinals, ordinals + n)));
    return 0;
  });
}
void keaki_host_device_free(void* dev) { delete (std::shared_ptr<Device>*)dev; }
size_t keaki_host_device_members(void* dev) { return (*(std::shared_ptr<Device>*)dev)->members(); }
// the keaki_hip_ctx of a member (member 0 for a single-GPU device): for the debug / option / memory calls of the C ABI
void* keaki_host_device_ctx(void* dev, size_t member) {
  auto& d = *(std::shared_ptr<Device>*)dev;
  return d->group() ? (void*)keaki_hip_group_ctx(d->group(), member) : (member == 0 ? (void*)d->ctx() : nullptr);
}
int keaki_host_setup_on(void* dev, const uint64_t* secret, size_t max_d, void** out) {
  return guard([&] {
    auto d = *(std::shared_ptr<Device>*)dev;
    *out = new Setup{d, kzg::KZGSetup::setup(d, fr_of(secret), max_d)};
    return 0;
  });
}
int keaki_host_setup_from_powers_on(void* dev, const uint64_t* g1_aff, size_t n, const uint64_t* tau_g2, void** out) {
  return guard([&] {
    auto d = *(std::shared_ptr<Device>*)dev;
    std::vector<G1> pts(n);
    for (size_t i = 0; i < n; i++) pts[i] = g1_of(g1_aff + 8 * i);
    *out = new Setup{d, kzg::KZGSetup::from_powers(d, std::move(pts), g2_of(tau_g2))};
    return 0;
  });
}

// KZGSetup::setup(secret, max_d)
int keaki_host_setup(int device, const uint64_t* secret, size_t max_d, void** out) {
  return guard([&] {
    auto dev = std::make_shared<Device>(device);
    auto* s = new Setup{dev, kzg::KZGSetup::setup(dev, fr_of(secret), max_d)};
    *out = s; return 0;
  });
}
int keaki_host_setup_from_powers(int device, const uint64_t* g1_aff, size_t n, const uint64_t* tau_g2, void** out) {
  return guard([&] {
    auto dev = std::make_shared<Device>(device);
    std::vector<G1> pts(n);
    for (size_t i = 0; i < n; i++) pts[i] = g1_of(g1_aff + 8 * i);
    *out = new Setup{dev, kzg::KZGSetup::from_powers(dev, std::move(pts), g2_of(tau_g2))};
    return 0;
  });
}
// ---- .ptau ingest (src/kzg/ptau.rs, src/kzg.rs:33-52). Status 2 = SetupFileError: kind / payload in err_out[3], text via last_error.
static int file_err(const ptau::SetupFileError& e, uint64_t* err_out) {
  if (err_out) { err_out[0] = (uint64_t)e.kind; err_out[1] = e.a; err_out[2] = e.b; }
  g_err = e.to_string();
  return 2;
}
// parse only (no GPU): header (modulus bytes up to mod_cap, power, ceremony power), section table, and the point counts; with
// g1_out / g2_out non-null the limbs of sections 2 / 3 are copied out (call once with nulls for the counts)
int keaki_host_ptau_parse(const char* path, uint8_t* modulus_out, size_t mod_cap, uint32_t* mod_len, uint32_t* power, uint32_t* ceremony_power,
                          uint64_t* sections_out /* 11 x (id, size, position) */, size_t* file_len, size_t* n_g1, size_t* n_g2, uint64_t* g1_out,
                          uint64_t* g2_out, uint64_t* err_out) {
  return guard([&] {
    auto data = ptau::load(path);
    if (!data.ok) return file_err(data.error, err_out);
    if (file_len) *file_len = data.value.size();
    auto md = ptau::verify_metadata(data.value);
    if (!md.ok) return file_err(md.error, err_out);
    auto secs = ptau::parse_sections(data.value);
    if (!secs.ok) return file_err(secs.error, err_out);
    if (sections_out) for (size_t i = 0; i < ptau::N_SECTIONS; i++) {
      sections_out[3 * i] = secs.value.sections[i].id; sections_out[3 * i + 1] = secs.value.sections[i].size; sections_out[3 * i + 2] = secs.value.sections[i].position;
    }
    auto hdr = ptau::parse_header(data.value, secs.value);
    if (!hdr.ok) return file_err(hdr.error, err_out);
    if (mod_len) *mod_len = (uint32_t)hdr.value.field_modulus.size();
    if (modulus_out) memcpy(modulus_out, hdr.value.field_modulus.data(), std::min(mod_cap, hdr.value.field_modulus.size()));
    if (power) *power = hdr.value.power;
    if (ceremony_power) *ceremony_power = hdr.value.ceremony_power;
    auto g1 = ptau::parse_tau_g1(data.value, secs.value, hdr.value.power);
    if (!g1.ok) return file_err(g1.error, err_out);
    auto g2 = ptau::parse_tau_g2(data.value, secs.value, hdr.value.power);
    if (!g2.ok) return file_err(g2.error, err_out);
    if (n_g1) *n_g1 = g1.value.size();
    if (n_g2) *n_g2 = g2.value.size();
    if (g1_out) for (size_t i = 0; i < g1.value.size(); i++) memcpy(g1_out + 8 * i, g1.value[i].w.data(), 64);
    if (g2_out) for (size_t i = 0; i < g2.value.size(); i++) memcpy(g2_out + 16 * i, g2.value[i].w.data(), 128);
    return 0;
  });
}
int keaki_host_ptau_section_info(const uint8_t* header12, size_t offset, uint64_t* out3, uint64_t* err_out) {
  return guard([&] {
    auto si = ptau::section_info_from_data(header12, offset);
    if (!si.ok) return file_err(si.error, err_out);
    out3[0] = si.value.id; out3[1] = si.value.size; out3[2] = si.value.position;
    return 0;
  });
}
int keaki_host_ptau_section_index(uint8_t id) { return ptau::section_index(id); }
// KZGSetup::new_from_file
int keaki_host_setup_from_file(int device, const char* path, void** out, uint64_t* err_out) {
  return guard([&] {
    auto dev = std::make_shared<Device>(device);
    auto r = kzg::new_from_file(dev, path);
    if (!r.ok) return file_err(r.error, err_out);
    *out = new Setup{dev, std::move(*r.setup)};
    return 0;
  });
}
void keaki_host_setup_free(void* s) { delete (Setup*)s; }
size_t keaki_host_setup_len(void* s) { return ((Setup*)s)->s.g1_pow().size(); }
void keaki_host_setup_g1_pow(void* s, size_t i, uint64_t* out) { memcpy(out, ((Setup*)s)->s.g1_pow()[i].w.data(), 64); }
void keaki_host_setup_g1_all(void* s, uint64_t* out) {   // all of g1_pow at once: len x 8 words
  const auto& v = ((Setup*)s)->s.g1_pow();
  for (size_t i = 0; i < v.size(); i++) memcpy(out + 8 * i, v[i].w.data(), 64);
}
void keaki_host_setup_tau_g2(void* s, uint64_t* out) { memcpy(out, ((Setup*)s)->s.tau_g2().w.data(), 128); }

int keaki_host_commit(void* s, const uint64_t* coeffs, size_t n, uint64_t* out_g1, uint64_t* err_out) {
  return guard([&] {
    auto r = kzg::commit(((Setup*)s)->s, frs_of(coeffs, n));
    if (!r.ok) return kzg_err(r.error, err_out);
    memcpy(out_g1, r.value.w.data(), 64); return 0;
  });
}
int keaki_host_open(void* s, const uint64_t* coeffs, size_t n, const uint64_t* point, uint64_t* out_g1, uint64_t* err_out) {
  return guard([&] {
    auto r = kzg::open(((Setup*)s)->s, frs_of(coeffs, n), fr_of(point));
    if (!r.ok) return kzg_err(r.error, err_out);
    memcpy(out_g1, r.value.w.data(), 64); return 0;
  });
}
int keaki_host_verify(void* s, const uint64_t* com, const uint64_t* point, const uint64_t* value, const uint64_t* proof, int* out_ok) {
  return guard([&] {
    auto r = kzg::verify(((Setup*)s)->s, g1_of(com), fr_of(point), fr_of(value), g1_of(proof));
    *out_ok = r.value ? 1 : 0; return 0;
  });
}
int keaki_host_precompute_open_fk(void* s, size_t domain_size) {
  return guard([&] { kzg::precompute_open_fk(((Setup*)s)->s, domain_size); return 0; });
}
int keaki_host_open_fk(void* s, const uint64_t* coeffs, size_t n, size_t domain_size, uint64_t* out_g1s, uint64_t* err_out) {
  return guard([&] {
    auto r = kzg::open_fk(((Setup*)s)->s, frs_of(coeffs, n), domain_size);
    if (!r.ok) return kzg_err(r.error, err_out);
    for (size_t i = 0; i < r.value.size(); i++) memcpy(out_g1s + 8 * i, r.value[i].w.data(), 64);
    return 0;
  });
}
int keaki_host_encapsulate(void* rng, void* s, const uint64_t* com, const uint64_t* point, const uint64_t* value, size_t msg_len,
                           uint64_t* ct_out, uint8_t* key_out) {
  return guard([&] {
    auto r = kem::encapsulate(*(Rng*)rng, ((Setup*)s)->s, g1_of(com), fr_of(point), fr_of(value), msg_len);
    memcpy(ct_out, r.first.w.data(), 128);
    if (msg_len) memcpy(key_out, r.second.data(), msg_len);
    return 0;
  });
}
int keaki_host_kem_prepare(void* s, size_t batch_hint) {
  return guard([&] { kem::prepare(((Setup*)s)->s, batch_hint); return 0; });
}
int keaki_host_decapsulate(void* s, const uint64_t* proof, const uint64_t* ct, size_t msg_len, uint8_t* key_out) {
  return guard([&] {
    auto k = kem::decapsulate(((Setup*)s)->s, g1_of(proof), g2_of(ct), msg_len);
    if (msg_len) memcpy(key_out, k.data(), msg_len);
    return 0;
  });
}
int keaki_host_encrypt(void* rng, void* s, const uint64_t* com, const uint64_t* point, const uint64_t* value, const uint8_t* msg, size_t len,
                       uint64_t* ct_g2_out, uint8_t* ct_msg_out) {
  return guard([&] {
    auto c = enc::encrypt(*(Rng*)rng, ((Setup*)s)->s, g1_of(com), fr_of(point), fr_of(value), std::vector<uint8_t>(msg, msg + len));
    memcpy(ct_g2_out, c.first.w.data(), 128);
    if (len) memcpy(ct_msg_out, c.second.data(), len);
    return 0;
  });
}
int keaki_host_decrypt(void* s, const uint64_t* proof, const uint64_t* ct_g2, const uint8_t* ct_msg, size_t len, uint8_t* msg_out) {
  return guard([&] {
    enc::Ciphertext c{g2_of(ct_g2), std::vector<uint8_t>(ct_msg, ct_msg + len)};
    auto m = enc::decrypt(((Setup*)s)->s, g1_of(proof), c);
    if (len) memcpy(msg_out, m.data(), len);
    return 0;
  });
}
// vec_commit: v has n scalars; outputs commitment and `domain_size` proofs (caller sizes proofs_out with keaki_host_domain(n+1))
int keaki_host_vec_commit(void* rng, void* s, const uint64_t* v, size_t n, uint64_t* com_out, uint64_t* proofs_out) {
  return guard([&] {
    static_assert(sizeof(Fr) == 32, "Fr is four u64 limbs");
    G1 com = vec::vec_commit_flat(*(Rng*)rng, ((Setup*)s)->s, reinterpret_cast<const Fr*>(v), n, proofs_out);      // straight into the caller's array
    memcpy(com_out, com.w.data(), 64);
    return 0;
  });
}
// vec_encrypt with equal-length messages (msg_len each, concatenated)
int keaki_host_vec_encrypt(void* rng, void* s, const uint64_t* com, const uint64_t* points, const uint64_t* values, const uint8_t* msgs,
                           size_t n, size_t msg_len, uint64_t* ct_g2_out, uint8_t* ct_msg_out) {
  return guard([&] {
    static_assert(sizeof(Fr) == 32, "Fr is four u64 limbs");
    vec::vec_encrypt_flat(*(Rng*)rng, ((Setup*)s)->s, g1_of(com), reinterpret_cast<const Fr*>(points), reinterpret_cast<const Fr*>(values), msgs, n, msg_len,
                          ct_g2_out, ct_msg_out);
    return 0;
  });
}
int keaki_host_vec_decrypt(void* s, const uint64_t* proofs, const uint64_t* ct_g2, const uint8_t* ct_msgs, size_t n, size_t msg_len,
                           uint8_t* msgs_out) {
  return guard([&] {
    vec::vec_decrypt_flat(((Setup*)s)->s, proofs, ct_g2, ct_msgs, n, msg_len, msgs_out);
    return 0;
  });
}

// ---- keaki::dist: one process per GPU (rank / world), the exchange of the 96-byte partials is the caller's ------------------------
int keaki_host_prepare_shard(void* s, size_t rank, size_t world) {
  return guard([&] { dist::prepare(((Setup*)s)->s, dist::Shard{rank, world}); return 0; });
}
int keaki_host_setup_has_tables(void* s) { return ((Setup*)s)->s.has_window_tables() ? 1 : 0; }
int keaki_host_commit_partial(void* s, const uint64_t* coeffs, size_t n, size_t rank, size_t world, uint64_t* out_jac12, uint64_t* err_out) {
  return guard([&] {
    auto r = dist::commit_partial(((Setup*)s)->s, frs_of(coeffs, n), dist::Shard{rank, world});
    if (!r.ok) return kzg_err(r.error, err_out);
    memcpy(out_jac12, r.value.data(), 96); return 0;
  });
}
// vec_commit with the commit left as this rank's partial; proofs_out sized with keaki_host_domain(n + 1)
int keaki_host_vec_commit_partial(void* rng, void* s, const uint64_t* v, size_t n, size_t rank, size_t world, uint64_t* out_jac12, uint64_t* proofs_out) {
  return guard([&] {
    auto r = dist::vec_commit_partial(*(Rng*)rng, ((Setup*)s)->s, frs_of(v, n), dist::Shard{rank, world});
    memcpy(out_jac12, r.first.data(), 96);
    for (size_t i = 0; i < r.second.size(); i++) memcpy(proofs_out + 8 * i, r.second[i].w.data(), 64);
    return 0;
  });
}
// ---- open_fk sharded (dist::ShardedOpenFk): the two callbacks are the caller's collectives over device memory
typedef int (*keaki_host_a2a_fn)(void* user, void* d_send, void* d_recv, size_t bytes_per_peer);
int keaki_host_fk_shard_can(void* s, size_t domain_size, size_t rank, size_t world) {
  return dist::ShardedOpenFk::can_shard(((Setup*)s)->s, domain_size, dist::Shard{rank, world}) ? 1 : 0;
}
int keaki_host_fk_shard_new(void* s, size_t domain_size, size_t rank, size_t world, void** out) {
  return guard([&] { *out = new dist::ShardedOpenFk(((Setup*)s)->s, domain_size, dist::Shard{rank, world}); return 0; });
}
void keaki_host_fk_shard_free(void* fk) { delete (dist::ShardedOpenFk*)fk; }
size_t keaki_host_fk_shard_buffer_bytes(void* fk) { return ((dist::ShardedOpenFk*)fk)->buffer_bytes(); }
int keaki_host_fk_shard_prepare(void* fk, void* d_send, void* d_recv, keaki_host_a2a_fn all_to_all, keaki_host_a2a_fn all_gather, void* user) {
  return guard([&] { ((dist::ShardedOpenFk*)fk)->prepare(d_send, d_recv, dist::FkExchange{all_to_all, all_gather, user}); return 0; });
}
int keaki_host_fk_shard_open(void* fk, const uint64_t* coeffs, size_t n, void* d_send, void* d_recv, keaki_host_a2a_fn all_to_all,
                             keaki_host_a2a_fn all_gather, void* user, uint64_t* out_g1s) {
  return guard([&] {
    auto r = ((dist::ShardedOpenFk*)fk)->open(frs_of(coeffs, n), d_send, d_recv, dist::FkExchange{all_to_all, all_gather, user});
    for (size_t i = 0; i < r.size(); i++) memcpy(out_g1s + 8 * i, r[i].w.data(), 64);
    return 0;
  });
}
int keaki_host_vec_commit_partial_fk(void* rng, void* s, const uint64_t* v, size_t n, size_t rank, size_t world, void* fk, void* d_send, void* d_recv,
                                     keaki_host_a2a_fn all_to_all, keaki_host_a2a_fn all_gather, void* user, uint64_t* out_jac12, uint64_t* proofs_out) {
  return guard([&] {
    auto r = dist::vec_commit_partial_fk(*(Rng*)rng, ((Setup*)s)->s, frs_of(v, n), dist::Shard{rank, world}, *(dist::ShardedOpenFk*)fk, d_send, d_recv,
                                         dist::FkExchange{all_to_all, all_gather, user});
    memcpy(out_jac12, r.first.data(), 96);
    for (size_t i = 0; i < r.second.size(); i++) memcpy(proofs_out + 8 * i, r.second[i].w.data(), 64);
    return 0;
  });
}
int keaki_host_commit_combine(void* s, const uint64_t* partials_jac12, size_t world, uint64_t* out_g1) {
  return guard([&] {
    static_assert(sizeof(dist::Partial) == 96, "a partial is twelve u64 words");
    G1 g = dist::commit_combine(((Setup*)s)->s, reinterpret_cast<const dist::Partial*>(partials_jac12), world);
    memcpy(out_g1, g.w.data(), 64); return 0;
  });
}
// vec_encrypt for the items of rank `rank` only (outputs sized for that rank's hi - lo items)
int keaki_host_vec_encrypt_shard(void* rng, void* s, const uint64_t* com, const uint64_t* points, const uint64_t* values, const uint8_t* msgs,
                                 size_t n, size_t msg_len, size_t rank, size_t world, uint64_t* ct_g2_out, uint8_t* ct_msg_out) {
  return guard([&] {
    dist::vec_encrypt_flat_shard(*(Rng*)rng, ((Setup*)s)->s, g1_of(com), reinterpret_cast<const Fr*>(points), reinterpret_cast<const Fr*>(values), msgs, n,
                                 msg_len, dist::Shard{rank, world}, ct_g2_out, ct_msg_out);
    return 0;
  });
}

}  // extern "C"
