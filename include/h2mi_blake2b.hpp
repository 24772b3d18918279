// h2mi_blake2b.hpp — Blake2b (RFC 7693) with a personalisation string, 64-byte digests, clonable state.
// Used by h2mi_transcript.hpp for the prover's Fiat-Shamir transcript (halo2_proofs::transcript uses the
// blake2b_simd crate: Params::new().hash_length(64).personal(b"Halo2-Transcript")).  Self-contained: no
// dependency on libh2mi, pinned against Python's hashlib in tests/test_formats.py.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <string>

namespace h2mi {
namespace blake2b {

class State {
 public:
  explicit State(const std::string& personal = std::string(), uint8_t digest_len = 64) : outlen_(digest_len) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                   0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    uint8_t param[64] = {0};
    param[0] = digest_len;  // digest length
    param[2] = 1;           // fanout
    param[3] = 1;           // depth
    std::memcpy(param + 48, personal.data(), personal.size() < 16 ? personal.size() : 16);
    for (int i = 0; i < 8; i++) {
      uint64_t w;
      std::memcpy(&w, param + 8 * i, 8);
      h_[i] = IV[i] ^ w;
    }
  }
  void update(const void* data, size_t len) {
    const uint8_t* p = static_cast<const uint8_t*>(data);
    while (len) {
      if (fill_ == 128) {  // the buffer holds a full block and more input follows: it is not the last one
        t_ += 128;
        compress(false);
        fill_ = 0;
      }
      size_t take = 128 - fill_ < len ? 128 - fill_ : len;
      std::memcpy(buf_ + fill_, p, take);
      fill_ += take;
      p += take;
      len -= take;
    }
  }
  void update(uint8_t byte) { update(&byte, 1); }
  // digest of everything absorbed so far; the state itself is left untouched (it can keep absorbing)
  std::array<uint8_t, 64> digest() const {
    State c = *this;
    c.t_ += c.fill_;
    std::memset(c.buf_ + c.fill_, 0, 128 - c.fill_);
    c.compress(true);
    std::array<uint8_t, 64> out{};
    std::memcpy(out.data(), c.h_, c.outlen_);
    return out;
  }

 private:
  static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
  void compress(bool last) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                   0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    static const uint8_t SIGMA[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t m[16], v[16];
    std::memcpy(m, buf_, 128);
    for (int i = 0; i < 8; i++) {
      v[i] = h_[i];
      v[i + 8] = IV[i];
    }
    v[12] ^= t_;  // low word of the byte counter (inputs beyond 2^64 bytes are out of scope)
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
      v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
      v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
      v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
      v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; r++) {
      const uint8_t* s = SIGMA[r];
      G(0, 4, 8, 12, m[s[0]], m[s[1]]);
      G(1, 5, 9, 13, m[s[2]], m[s[3]]);
      G(2, 6, 10, 14, m[s[4]], m[s[5]]);
      G(3, 7, 11, 15, m[s[6]], m[s[7]]);
      G(0, 5, 10, 15, m[s[8]], m[s[9]]);
      G(1, 6, 11, 12, m[s[10]], m[s[11]]);
      G(2, 7, 8, 13, m[s[12]], m[s[13]]);
      G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) h_[i] ^= v[i] ^ v[i + 8];
  }
  uint64_t h_[8];
  uint64_t t_ = 0;
  uint8_t buf_[128] = {0};
  size_t fill_ = 0;
  uint8_t outlen_;
};

}  // namespace blake2b
}  // namespace h2mi
