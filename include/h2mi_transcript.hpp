// h2mi_transcript.hpp — C++ mirror of halo2_proofs::transcript::{Blake2bWrite, Blake2bRead, Challenge255}
// for bn256::G1Affine (reference call sites: examples/standard_plonk.rs:40-49,56; src/scaffold.rs:190-199).
//
// Restated [RECALL halo2_proofs v2023_02_02 transcript.rs + halo2curves 0.3.x; the crates are not available
// here — see csrc/h2mi_serde.hip for the encoding conventions]: Blake2b-512, personalisation
// "Halo2-Transcript"; prefix bytes 0 = challenge, 1 = point, 2 = scalar; a point is absorbed as
// x.to_repr() || y.to_repr() and written compressed (32 bytes); a scalar is absorbed and written as
// to_repr(); a challenge is the digest of a clone of the state reduced mod r (Fr::from_bytes_wide).
// Host control plane: tens of values per proof, no device work except decompression on the read side.
#pragma once
#include <array>
#include <vector>

#include "h2mi.hpp"
#include "h2mi_blake2b.hpp"

namespace h2mi {
namespace serde {

constexpr uint64_t FQ_MODULUS[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
constexpr uint64_t FQ_INV = 0x87d20782e4866389ULL;

// one Montgomery reduction: a * 2^-256 mod m (Montgomery limbs -> canonical integer)
inline void from_mont(const uint64_t a[4], const uint64_t mod[4], uint64_t inv, uint64_t out[4]) {
  typedef unsigned __int128 u128;
  uint64_t t[5] = {a[0], a[1], a[2], a[3], 0};
  for (int i = 0; i < 4; i++) {
    uint64_t m = t[0] * inv;
    u128 c = (u128)m * mod[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * mod[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = (uint64_t)(c >> 64);
  }
  bool ge = t[4] != 0;
  if (!ge) {
    ge = true;
    for (int i = 3; i >= 0; i--) {
      if (t[i] > mod[i]) break;
      if (t[i] < mod[i]) { ge = false; break; }
    }
  }
  if (ge) {
    u128 bo = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)t[i] - mod[i] - (uint64_t)bo;
      t[i] = (uint64_t)d;
      bo = (d >> 64) & 1;
    }
  }
  for (int i = 0; i < 4; i++) out[i] = t[i];
}
typedef std::array<uint8_t, 32> Repr;
inline Repr limbs_to_bytes(const uint64_t l[4]) {  // little-endian host
  Repr r;
  std::memcpy(r.data(), l, 32);
  return r;
}
// Fr::to_repr
inline Repr fr_to_repr(const Fr& a) {
  uint64_t c[4];
  from_mont(a.l, fr::MODULUS, fr::INV, c);
  return limbs_to_bytes(c);
}
// Fr::from_repr: throws on a non-canonical encoding (the crate returns CtOption::none)
inline Fr fr_from_repr(const Repr& b) {
  Fr raw;
  std::memcpy(raw.l, b.data(), 32);
  for (int i = 3; i >= 0; i--) {
    if (raw.l[i] < fr::MODULUS[i]) return fr::mul(raw, fr::R2);
    if (raw.l[i] > fr::MODULUS[i]) break;
  }
  throw Error(H2MI_EINVAL, "Fr::from_repr: not canonical");
}
// Fr::from_bytes_wide: 64 little-endian bytes reduced mod r
inline Fr fr_from_bytes_wide(const std::array<uint8_t, 64>& b) {
  Fr lo, hi;
  std::memcpy(lo.l, b.data(), 32);
  std::memcpy(hi.l, b.data() + 32, 32);
  Fr lo_m = fr::mul(lo, fr::R2);                 // lo * R
  Fr hi_m = fr::mul(fr::mul(hi, fr::R2), fr::R2);  // hi * 2^256 * R
  return fr::add(lo_m, hi_m);
}
// G1Affine::to_bytes: x with flags in byte 31 (0x40: y odd, 0x80: identity)
inline Repr g1_to_bytes(const G1Affine& p) {
  bool ident = true;
  for (int i = 0; i < 4; i++) ident = ident && p.x[i] == 0 && p.y[i] == 0;
  Repr r{};
  if (ident) {
    r[31] = 0x80;
    return r;
  }
  uint64_t x[4], y[4];
  from_mont(p.x, FQ_MODULUS, FQ_INV, x);
  from_mont(p.y, FQ_MODULUS, FQ_INV, y);
  r = limbs_to_bytes(x);
  if (y[0] & 1) r[31] |= 0x40;
  return r;
}
// G1Affine::from_bytes (square root on the device); throws on an invalid encoding
inline G1Affine g1_from_bytes(const Repr& b) {
  G1Affine p;
  uint64_t bad = 0;
  check(h2mi_g1_decompress(b.data(), 1, (uint64_t*)&p, &bad), "g1_decompress");
  if (bad) throw Error(H2MI_EINVAL, "G1Affine::from_bytes: invalid encoding");
  return p;
}

}  // namespace serde

namespace transcript {

class Blake2bBase {
 public:
  Blake2bBase() : state_("Halo2-Transcript") {}
  // Challenge255: squeeze_challenge().get_scalar()
  Fr squeeze_challenge() {
    state_.update((uint8_t)0);
    return serde::fr_from_bytes_wide(state_.digest());
  }
  void common_point(const G1Affine& p) {
    bool ident = true;
    for (int i = 0; i < 4; i++) ident = ident && p.x[i] == 0 && p.y[i] == 0;
    if (ident) throw Error(H2MI_EINVAL, "cannot write points at infinity to the transcript");
    uint64_t x[4], y[4];
    serde::from_mont(p.x, serde::FQ_MODULUS, serde::FQ_INV, x);
    serde::from_mont(p.y, serde::FQ_MODULUS, serde::FQ_INV, y);
    state_.update((uint8_t)1);
    state_.update(x, 32);
    state_.update(y, 32);
  }
  void common_scalar(const Fr& s) {
    state_.update((uint8_t)2);
    state_.update(serde::fr_to_repr(s).data(), 32);
  }

 protected:
  blake2b::State state_;
};

class Blake2bWrite : public Blake2bBase {
 public:
  // TranscriptWriterBuffer::init — the reference writes Blake2bWrite::<_, _, Challenge255<_>>::init(vec![])
  static Blake2bWrite init() { return Blake2bWrite(); }
  void write_point(const G1Affine& p) {
    common_point(p);
    auto b = serde::g1_to_bytes(p);
    proof_.insert(proof_.end(), b.begin(), b.end());
  }
  void write_scalar(const Fr& s) {
    common_scalar(s);
    auto b = serde::fr_to_repr(s);
    proof_.insert(proof_.end(), b.begin(), b.end());
  }
  const std::vector<uint8_t>& finalize() const { return proof_; }

 private:
  std::vector<uint8_t> proof_;
};

class Blake2bRead : public Blake2bBase {
 public:
  explicit Blake2bRead(std::vector<uint8_t> proof) : proof_(std::move(proof)) {}
  // TranscriptReadBuffer::init — Blake2bRead::<_, _, Challenge255<_>>::init(&proof[..])
  static Blake2bRead init(std::vector<uint8_t> proof) { return Blake2bRead(std::move(proof)); }
  G1Affine read_point() {
    G1Affine p = serde::g1_from_bytes(take());
    common_point(p);
    return p;
  }
  Fr read_scalar() {
    Fr s = serde::fr_from_repr(take());
    common_scalar(s);
    return s;
  }

 private:
  serde::Repr take() {
    if (pos_ + 32 > proof_.size()) throw Error(H2MI_EINVAL, "transcript: proof too short");
    serde::Repr r;
    std::memcpy(r.data(), proof_.data() + pos_, 32);
    pos_ += 32;
    return r;
  }
  std::vector<uint8_t> proof_;
  size_t pos_ = 0;
};

}  // namespace transcript
}  // namespace h2mi
