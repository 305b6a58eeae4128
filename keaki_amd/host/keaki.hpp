// keaki.hpp -- C++ mirror of keaki's public Rust API for E = Bn254, above the C ABI of
// libkeaki_hip.so. (The reference's own toolchain, Rust, is absent from this image; INTEGRATION.md
// shows the feature-gated Rust shim that binds the same C ABI.)
//
// Same module / function names, argument meaning and error behaviour as the reference:
//   keaki::kzg::KZGSetup, commit, open, verify, KZGError::PolynomialTooLarge   (src/kzg.rs:22-151,205-209)
//   keaki::kem::encapsulate, decapsulate                                        (src/kem.rs:13-72)
//   keaki::enc::Ciphertext, encrypt, decrypt                                    (src/enc.rs:13-55)
//   keaki::vec::PADDING_LEN, vec_commit, vec_encrypt, vec_decrypt               (src/vec.rs:18-81)
// All group / pairing arithmetic runs on the GPU through the C ABI; the host only does what keaki's
// own glue does on the host: O(n) scalar-field work (Horner, quotient, iFFT), RNG draws, XOR.
// There is no CPU fallback for the group arithmetic.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/keaki_hip.h"

namespace keaki {

// ---- scalar field Fr (ark-ff Fp<MontBackend<FrConfig,4>>): 4 x u64 Montgomery limbs ------------------
struct Fr {
  uint64_t l[4] = {0, 0, 0, 0};
  static Fr zero() { return Fr(); }
  static Fr one();
  static Fr from_u64(uint64_t v);
  static Fr from_i64(int64_t v);  // Fr::from(-24) in the reference tests
  bool operator==(const Fr& o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
  bool operator!=(const Fr& o) const { return !(*this == o); }
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  Fr operator+(const Fr& o) const;
  Fr operator-(const Fr& o) const;
  Fr operator*(const Fr& o) const;
  Fr operator-() const;
  Fr pow(uint64_t e) const;
  Fr inverse() const;  // Fermat; zero -> zero
};

// rand::Rng: the only thing keaki needs from it is next_u64 (Fr::rand draws 4 per candidate)
struct Rng {
  virtual ~Rng() {}
  virtual uint64_t next_u64() = 0;
};
// ark-ff 0.4.2 `impl Distribution<Fp> for Standard`: 4 x next_u64, clear the top 2 bits, reject >= r;
// the accepted limbs ARE the Montgomery representation (src/kem.rs:26, src/vec.rs:32).
Fr fr_rand(Rng& rng);

// ---- group elements: affine, Montgomery limbs, identity = all zero (the C-ABI encodings) --------------
struct G1 {
  std::array<uint64_t, 8> w{};
  bool operator==(const G1& o) const { return w == o.w; }
  bool operator!=(const G1& o) const { return w != o.w; }
  bool is_zero() const { for (auto x : w) if (x) return false; return true; }
};
struct G2 {
  std::array<uint64_t, 16> w{};
  bool operator==(const G2& o) const { return w == o.w; }
  bool operator!=(const G2& o) const { return w != o.w; }
};

struct HipError : std::runtime_error {
  int status;
  HipError(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

// One GPU context shared by the setup objects created from it -- or, built from a LIST of ordinals, several GPUs of this process behind
// one object (keaki_hip_group_*: one context and one host thread per entry inside the library; an ordinal may repeat). With a group,
// kzg::commit / open spread the MSM over all members by SRS range (each member keeps its chunk of the SRS and that chunk's window
// tables from setup on; the 96-byte partial sums come back through host memory) and the loops of vec_encrypt / vec_decrypt are
// split by item range. Everything else (verify, single encapsulate / decapsulate, open_fk) runs on member 0, whose context ctx() returns.
class Device {
 public:
  explicit Device(int ordinal = 0);
  explicit Device(const std::vector<int>& ordinals);
  ~Device();
  Device(const Device&) = delete;
  keaki_hip_ctx* ctx() const { return ctx_; }
  keaki_hip_group* group() const { return group_; }   // null for a single-GPU device
  size_t members() const;
  void check(int status) const;        // throws HipError
  void check_group(int status) const;  // the same for keaki_hip_group_* calls
 private:
  keaki_hip_ctx* ctx_ = nullptr;
  keaki_hip_group* group_ = nullptr;
};
// batches below this many items stay on member 0 of a group (a launch per member costs more than it saves)
constexpr size_t GROUP_MIN_ITEMS = 1024;

using DensePolynomial = std::vector<Fr>;  // coefficients, low degree first (ark-poly DensePolynomial::coeffs)

namespace kzg {

// src/kzg.rs:205-209
struct KZGError {
  enum Kind { PolynomialTooLarge } kind;
  size_t degree, max_degree;  // "{0}" and "{1}" of the thiserror message
  bool operator==(const KZGError& o) const { return kind == o.kind && degree == o.degree && max_degree == o.max_degree; }
  std::string to_string() const;
};
template <class T>
struct Result {  // Result<T, KZGError>
  bool ok;
  T value;
  KZGError error;
  static Result Ok(T v) { return Result{true, std::move(v), KZGError{KZGError::PolynomialTooLarge, 0, 0}}; }
  static Result Err(KZGError e) { return Result{false, T(), e}; }
  const T& unwrap() const { if (!ok) throw std::runtime_error("called unwrap() on Err: " + error.to_string()); return value; }
};

// src/kzg.rs:22-86. g1_pow and g1_aff hold the same affine values here (the reference keeps a
// projective and an affine copy; equality and serialisation only ever see the affine value).
class KZGSetup {
 public:
  // `setup(secret, max_d)`: [secret^i]_1 for i < max_d and [secret]_2.  "Don't use this." (src/kzg.rs:54)
  static KZGSetup setup(std::shared_ptr<Device> dev, const Fr& secret, size_t max_d);
  // from already-known powers (what new_from_file produces after parsing; the .ptau parser itself is out of scope)
  static KZGSetup from_powers(std::shared_ptr<Device> dev, std::vector<G1> g1_aff, const G2& tau_g2);
  ~KZGSetup();
  KZGSetup(KZGSetup&&) noexcept;
  KZGSetup(const KZGSetup&) = delete;
  const std::vector<G1>& g1_pow() const { return g1_aff_; }
  const std::vector<G1>& g1_aff() const { return g1_aff_; }
  const G2& tau_g2() const { return tau_g2_; }
  const std::shared_ptr<Device>& device() const { return dev_; }
  // device-resident copy of the whole SRS on (member 0 of) the device. With a group it is uploaded on first use: commit / open never
  // need it (they read the per-member chunks), open_fk and the keaki::dist calls do.
  keaki_hip_srs_g1* srs() const;
  keaki_hip_group_srs_g1* group_srs() const { return gsrs_; }
  // group devices: the FK23 handle of domain size d = 2^log2d (every member's copy of srs[0..d) and its part of the SRS-only transform),
  // built on first use and kept until another size is asked for
  keaki_hip_group_fk* group_fk(unsigned log2d) const;
  // false when the optional window tables of the SRS did not fit in HBM: commit / open then run the generic per-window MSM (same results)
  bool has_window_tables() const { return tables_; }
  // this rank's chunk [lo, hi) of the SRS as a handle of its own, with its own window tables (built on first use; see keaki::dist)
  keaki_hip_srs_g1* chunk_srs(size_t lo, size_t hi) const;
 private:
  KZGSetup() {}
  std::shared_ptr<Device> dev_;
  std::vector<G1> g1_aff_;
  G2 tau_g2_;
  mutable keaki_hip_srs_g1* srs_ = nullptr;  // device-resident copy of g1_aff, uploaded once
  keaki_hip_group_srs_g1* gsrs_ = nullptr;   // group devices: one chunk (+ its window tables) per member
  mutable keaki_hip_group_fk* gfk_ = nullptr;
  mutable unsigned gfk_log2d_ = 0;
  bool tables_ = false;
  mutable keaki_hip_srs_g1* chunk_ = nullptr;
  mutable size_t chunk_lo_ = 0, chunk_hi_ = 0;
};

Result<G1> commit(const KZGSetup& setup, const DensePolynomial& p);                          // src/kzg.rs:89-101
Result<G1> open(const KZGSetup& setup, const DensePolynomial& p, const Fr& point);           // src/kzg.rs:104-124
Result<bool> verify(const KZGSetup& setup, const G1& commitment, const Fr& point, const Fr& value, const G1& proof);  // :127-151
// all openings at the roots of unity of a size-d domain (src/kzg.rs:157-203): FK23 -- three G1 FFTs + 2d scalar-mults on
// the GPU (keaki_hip_open_fk) when p.size() == domain_size is a power of two; per-point `open` otherwise.
Result<std::vector<G1>> open_fk(const KZGSetup& setup, const std::vector<Fr>& p, size_t domain_size);
// setup-time: tabulate the SRS-only part of open_fk (hat_s) for power-of-two domain size d, so the first open_fk does not pay for it
void precompute_open_fk(const KZGSetup& setup, size_t domain_size);

}  // namespace kzg

namespace kem {
// src/kem.rs:13-50 / :55-72
std::pair<G2, std::vector<uint8_t>> encapsulate(Rng& rng, const kzg::KZGSetup& setup, const G1& commitment, const Fr& point,
                                                const Fr& value, size_t msg_len);
std::vector<uint8_t> decapsulate(const kzg::KZGSetup& setup, const G1& proof, const G2& ciphertext, size_t msg_len);
// setup-time (optional): the tables of encapsulation that depend on the setup only (generators, [tau]_2, e(g1, g2)), for batches of
// `batch_hint` items per call -- on every member of a group device for its share. Like the SRS window tables: results never depend on it.
void prepare(const kzg::KZGSetup& setup, size_t batch_hint);
}  // namespace kem

namespace enc {
using Ciphertext = std::pair<G2, std::vector<uint8_t>>;  // src/enc.rs:13
Ciphertext encrypt(Rng& rng, const kzg::KZGSetup& setup, const G1& com, const Fr& point, const Fr& value, const std::vector<uint8_t>& msg);
std::vector<uint8_t> decrypt(const kzg::KZGSetup& setup, const G1& proof, const Ciphertext& ct);
}  // namespace enc

namespace vec {
constexpr size_t PADDING_LEN = 1;  // src/vec.rs:18
// Radix2EvaluationDomain over Fr (ark-poly): size = next power of two >= n, generator of that order
struct Radix2Domain {
  size_t size; Fr group_gen, group_gen_inv, size_inv;
  static Radix2Domain create(size_t min_size);
  std::vector<Fr> elements() const;
  std::vector<Fr> ifft(std::vector<Fr> evals) const;  // pads with zeros to `size`
  std::vector<Fr> fft(std::vector<Fr> coeffs) const;
};
// src/vec.rs:22-49
std::pair<G1, std::vector<G1>> vec_commit(Rng& rng, const kzg::KZGSetup& setup, const std::vector<Fr>& v);
// the same on contiguous buffers: proofs_out = domain_size affine points (8 words each). For domains of >= 2^12 evaluations on a single-GPU
// device everything behind the padding draw is ONE device call (keaki_hip_vec_commit: iFFT, FK23 openings, commit; the coefficients never
// return to the host); smaller domains and group devices take the steps one by one. Same values either way.
G1 vec_commit_flat(Rng& rng, const kzg::KZGSetup& setup, const Fr* v, size_t n, uint64_t* proofs_out);
// lines :27-37 of it (padding draw, iFFT): the coefficient vector over the whole domain, not trimmed
std::vector<Fr> vec_commit_coeffs(Rng& rng, const kzg::KZGSetup& setup, const std::vector<Fr>& v);
// lines :27-44 of it (padding draw, iFFT, open_fk): the dense coefficient vector and the proofs, without the final commit
std::pair<DensePolynomial, std::vector<G1>> vec_commit_openings(Rng& rng, const kzg::KZGSetup& setup, const std::vector<Fr>& v);
// src/vec.rs:52-69: one Fr::rand per item in index order, then ONE batched GPU call for all items
std::vector<enc::Ciphertext> vec_encrypt(Rng& rng, const kzg::KZGSetup& setup, const G1& com, const std::vector<Fr>& points,
                                         const std::vector<Fr>& values, const std::vector<std::vector<uint8_t>>& messages);
// src/vec.rs:72-81
void vec_encrypt_flat(Rng& rng, const kzg::KZGSetup& setup, const G1& com, const Fr* points, const Fr* values, const uint8_t* msgs, size_t n,
                      size_t msg_len, uint64_t* ct_g2_out, uint8_t* ct_msg_out);
void vec_decrypt_flat(const kzg::KZGSetup& setup, const uint64_t* proofs, const uint64_t* ct_g2, const uint8_t* ct_msgs, size_t n, size_t msg_len,
                      uint8_t* msgs_out);
std::vector<std::vector<uint8_t>> vec_decrypt(const kzg::KZGSetup& setup, const std::vector<G1>& proofs,
                                              const std::vector<const enc::Ciphertext*>& cts);
}  // namespace vec

// ---- one process per GPU: the same calls with the work split over `world` ranks ------------------------------------------------------
// commit's MSM (src/kzg.rs:98) shards by contiguous coefficient / point range: every rank runs a complete Pippenger on its range and
// emits one 96-byte normalised-Jacobian partial; the CALLER exchanges the partials (RCCL all-gather in the harness, any transport
// in an application: the library never opens a connection) and every rank adds them. EC addition is exact, so the affine result equals
// the single-GPU one bit for bit. The loops of vec_encrypt / vec_decrypt (src/vec.rs:63-66, :75-78) shard by item, no exchange at all.
namespace dist {
struct Shard {
  size_t rank = 0, world = 1;
  // contiguous range of rank `rank` out of n units; sizes differ by at most one
  std::pair<size_t, size_t> bounds(size_t n) const;
};
using Partial = std::array<uint64_t, 12>;   // normalised Jacobian (x, y, 1) or (1, 1, 0)
// setup-time: this rank's SRS chunk as a handle of its own with its window tables (commit_partial would build them on first use)
void prepare(const kzg::KZGSetup& setup, const Shard& sh);
// this rank's share of kzg::commit: sum over i in bounds(setup.len) and i < p.size() of p[i] [tau^i]_1. Same error as commit.
kzg::Result<Partial> commit_partial(const kzg::KZGSetup& setup, const DensePolynomial& p, const Shard& sh);
// vec_commit with the final commit (src/vec.rs:46) left as this rank's partial. Padding draw, iFFT and the FK23 openings are REPLICATED:
// every rank runs them with the same rng stream and gets the same proofs (vec_commit_partial_fk shards the openings too).
std::pair<Partial, std::vector<G1>> vec_commit_partial(Rng& rng, const kzg::KZGSetup& setup, const std::vector<Fr>& v, const Shard& sh);
// ---- kzg::open_fk (FK23, src/kzg.rs:157-203) sharded over the ranks: every rank does 1/world of the butterflies of the group FFTs (one
// inverse and one forward transform of size d: csrc/fft_g1.hip) and of the 2d scalar-mults (keaki_hip_fk_shard_*). Between the steps the
// ranks exchange 96-byte points: two all-to-alls of d / world^2 points per peer and one all-gather of the d / world affine proofs per
// rank and call, one all-to-all of 2d / world^2 points per peer at setup -- the one place of the whole path where xGMI bandwidth matters.
// The exchanges are the CALLER's (RCCL in an application: the library never opens a connection):
struct FkExchange {
  // d_send: `world` chunks of bytes_per_peer, chunk q for rank q; d_recv: the chunks received, in rank order. Device memory.
  // Must not return before d_recv is complete (the steps run on the Device's own stream). Returns 0, or non-zero to abort the call
  // (the function then throws std::runtime_error).
  int (*all_to_all)(void* user, void* d_send, void* d_recv, size_t bytes_per_peer);
  // d_send: bytes_per_rank of this rank; d_recv: every rank's, in rank order
  int (*all_gather)(void* user, void* d_send, void* d_recv, size_t bytes_per_rank);
  void* user;
};
class ShardedOpenFk {
 public:
  // domain_size: a power of two >= world^2 and <= the SRS; world a power of two >= 2 (can_shard). The setup must outlive every call of prepare / open.
  ShardedOpenFk(const kzg::KZGSetup& setup, size_t domain_size, const Shard& sh);
  ~ShardedOpenFk();
  ShardedOpenFk(const ShardedOpenFk&) = delete;
  static bool can_shard(const kzg::KZGSetup& setup, size_t domain_size, const Shard& sh);
  size_t buffer_bytes() const { return sizes_[0]; }       // of d_send and of d_recv (device memory, the caller's)
  size_t domain_size() const { return d_; }
  // setup time: this rank's part of hat_s = DFT_2d(reversed SRS); one all-to-all
  void prepare(void* d_send, void* d_recv, const FkExchange& ex);
  // == kzg::open_fk(setup, p, domain_size).unwrap() on every rank (p.size() == domain_size)
  std::vector<G1> open(const std::vector<Fr>& p, void* d_send, void* d_recv, const FkExchange& ex);
 private:
  const kzg::KZGSetup& setup_;
  std::shared_ptr<Device> dev_;           // keeps the context alive for the destructor even if the setup goes first
  keaki_hip_fk_shard* fk_ = nullptr;
  size_t d_ = 0, sizes_[4] = {0, 0, 0, 0};
  bool prepared_ = false;
};
// vec_commit_partial with the FK23 openings sharded as well: nothing but the padding draw and the scalar-field iFFT is replicated
std::pair<Partial, std::vector<G1>> vec_commit_partial_fk(Rng& rng, const kzg::KZGSetup& setup, const std::vector<Fr>& v, const Shard& sh,
                                                          ShardedOpenFk& fk, void* d_send, void* d_recv, const FkExchange& ex);
// the sum of all ranks' partials (in any order)
G1 commit_combine(const kzg::KZGSetup& setup, const Partial* partials, size_t world);
// vec_encrypt_flat for the items in bounds(n) only: draws ALL n values of r in index order (so every rank's stream, and therefore every
// ciphertext, equals the single-process call's) and writes (hi - lo) ciphertexts
void vec_encrypt_flat_shard(Rng& rng, const kzg::KZGSetup& setup, const G1& com, const Fr* points, const Fr* values, const uint8_t* msgs, size_t n,
                            size_t msg_len, const Shard& sh, uint64_t* ct_g2_out, uint8_t* ct_msg_out);
}  // namespace dist

}  // namespace keaki
