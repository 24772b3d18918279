// prints the 64-byte Blake2b digest (personalisation "Halo2-Transcript") of the hex string in argv[1];
// fed in uneven pieces so that the buffering paths of State::update are exercised.
#include <cstdio>
#include <string>
#include <vector>

#include "h2mi_blake2b.hpp"

int main(int argc, char** argv) {
  std::string hex = argc > 1 ? argv[1] : "";
  std::vector<unsigned char> msg;
  for (size_t i = 0; i + 1 < hex.size(); i += 2) msg.push_back((unsigned char)std::stoi(hex.substr(i, 2), nullptr, 16));
  h2mi::blake2b::State st("Halo2-Transcript");
  size_t pos = 0, step = 1;
  while (pos < msg.size()) {
    size_t take = step < msg.size() - pos ? step : msg.size() - pos;
    st.update(msg.data() + pos, take);
    pos += take;
    step = step * 3 + 1;
  }
  auto mid = st.digest();  // digest() must not disturb the running state
  (void)mid;
  auto d = st.digest();
  for (unsigned char b : d) std::printf("%02x", b);
  std::printf("\n");
  return 0;
}
