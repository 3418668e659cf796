// TEST INFRASTRUCTURE ONLY.  C-callable wrapper (our code) around the reference's own
// VLC.cpp, which is compiled unmodified from /root/reference by oracle/Makefile.
// It drives the reference exactly like Slices.cpp:469-612 drives it for one HQ slice
// component: bounded(8*bytes) << SignedVLC... << flush << align.
#include <cstdint>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>

#include "VLC.h"

extern "C" {

// returns bytes written, -1 on "Attempt to write beyond end of bounded write", -2 if cap too small
long ref_svlc_write_bounded(const int32_t *vals, int n, long bound_bits, uint8_t *out, long cap) {
  std::ostringstream ss;
  try {
    ss << vlc::bounded((int)bound_bits);
    for (int i = 0; i < n; ++i) ss << SignedVLC(vals[i]);
    ss << vlc::flush << vlc::align;
  } catch (const std::length_error &) {
    return -1;
  }
  const std::string s = ss.str();
  if ((long)s.size() > cap) return -2;
  std::memcpy(out, s.data(), s.size());
  return (long)s.size();
}

// reads n signed values from a bounded region; returns bytes consumed
long ref_svlc_read_bounded(const uint8_t *in, long len, long bound_bits, int n, int32_t *vals) {
  std::istringstream ss(std::string(reinterpret_cast<const char *>(in), (size_t)len));
  ss >> vlc::bounded((int)bound_bits);
  SignedVLC v;
  for (int i = 0; i < n; ++i) {
    ss >> v;
    vals[i] = v;
  }
  ss >> vlc::flush >> vlc::align;
  return (long)ss.tellg();
}

int ref_svlc_numbits(int32_t value) { return (int)SignedVLC(value).numOfBits(); }
unsigned ref_svlc_code(int32_t value) { return SignedVLC(value).code(); }
int ref_uvlc_numbits(uint32_t value) { return (int)UnsignedVLC(value).numOfBits(); }
unsigned ref_uvlc_code(uint32_t value) { return UnsignedVLC(value).code(); }
}
