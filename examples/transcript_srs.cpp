// C++ host layer, wire formats (include/h2mi_transcript.hpp, ParamsKZG::read / write): what the reference's
// examples do around create_proof — Blake2bWrite::init + finalize, Blake2bRead::init for verify_proof
// (reference examples/standard_plonk.rs:40-49,56) and the SRS file cached by gen_srs-style helpers.
//
// Usage: transcript_srs <values.bin> [<srs_in> <srs_out>]
//   values.bin: u32 n_points, u32 n_scalars, n_points x 64 B G1Affine, n_scalars x 32 B Fr (in-memory layouts)
// Writes all points, squeezes a challenge, writes all scalars, squeezes again; prints the proof and the two
// challenges in hex; reads the proof back with Blake2bRead and checks points, scalars and challenges.
// With an SRS file: ParamsKZG::read, one commitment as a smoke test, ParamsKZG::write to <srs_out>.
#include <cstdio>
#include <fstream>
#include <vector>

#include "h2mi_transcript.hpp"

using namespace h2mi;

static void print_hex(const char* tag, const void* p, size_t n) {
  std::printf("%s ", tag);
  for (size_t i = 0; i < n; i++) std::printf("%02x", ((const unsigned char*)p)[i]);
  std::printf("\n");
}

int main(int argc, char** argv) {
  if (argc < 2) return 64;
  try {
    init();
    std::ifstream in(argv[1], std::ios::binary);
    uint32_t np = 0, ns = 0;
    in.read((char*)&np, 4);
    in.read((char*)&ns, 4);
    std::vector<G1Affine> pts(np);
    std::vector<Fr> sc(ns);
    in.read((char*)pts.data(), (std::streamsize)np * 64);
    in.read((char*)sc.data(), (std::streamsize)ns * 32);
    if (!in) return 65;

    auto tw = transcript::Blake2bWrite::init();
    for (auto& p : pts) tw.write_point(p);
    Fr c1 = tw.squeeze_challenge();
    for (auto& s : sc) tw.write_scalar(s);
    Fr c2 = tw.squeeze_challenge();
    const std::vector<uint8_t>& proof = tw.finalize();
    print_hex("proof", proof.data(), proof.size());
    print_hex("challenge1", c1.l, 32);
    print_hex("challenge2", c2.l, 32);

    auto tr = transcript::Blake2bRead::init(proof);
    bool ok = true;
    for (auto& p : pts) {
      G1Affine q = tr.read_point();
      ok = ok && std::memcmp(&q, &p, 64) == 0;
    }
    ok = ok && tr.squeeze_challenge() == c1;
    for (auto& s : sc) ok = ok && tr.read_scalar() == s;
    ok = ok && tr.squeeze_challenge() == c2;
    bool threw = false;
    try {
      (void)tr.read_scalar();
    } catch (const Error&) {
      threw = true;
    }
    std::printf("transcript round trip %s\n", ok && threw ? "ok" : "FAILED");
    if (!(ok && threw)) return 1;

    if (argc >= 4) {
      std::ifstream srs(argv[2], std::ios::binary);
      auto params = poly::kzg::ParamsKZG::read(srs);
      std::vector<Fr> ones(params.n(), fr::ONE);
      G1 c = params.commit(ones);
      std::printf("srs k=%u commit(all ones) identity=%d\n", params.k(), (int)c.is_identity());
      std::ofstream out(argv[3], std::ios::binary);
      params.write(out);
    }
    h2mi_shutdown();
    return 0;
  } catch (const Error& e) {
    std::fprintf(stderr, "h2mi error %d: %s\n", e.code, e.what());
    return 2;
  }
}
