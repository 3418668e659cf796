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

// One LD slice written the way LDSliceIO writes it (Slices.cpp:195-244), by the reference's own Bits / SignedVLC /
// bounded / flush / align: 7-bit quantiser index, the luma bit count in uv_split_bits bits, the luma codes bounded by
// y_bits, the chroma codes (u and v alternating: uv holds them interleaved) bounded by the remainder, zero padding.
// The caller supplies the coefficient order and the two counts (luma_slice_bits, utils::intlog2: Slices.cpp:51-68,
// Utils.cpp:40-48).  Returns bytes written, -1 on a bounded-write overflow.
long ref_ld_slice_write(int q_index, int slice_bytes, int uv_split_bits, int y_bits, const int32_t *y, int ny,
                        const int32_t *uv, int nuv, uint8_t *out, long cap) {
  std::ostringstream ss;
  try {
    ss << Bits(7, (unsigned)q_index);
    const int uv_bits = 8 * slice_bytes - 7 - uv_split_bits - y_bits;
    ss << Bits((unsigned)uv_split_bits, (unsigned)y_bits);
    ss << vlc::bounded(y_bits);
    for (int i = 0; i < ny; ++i) ss << SignedVLC(y[i]);
    ss << vlc::flush;
    ss << vlc::bounded(uv_bits);
    for (int i = 0; i < nuv; ++i) ss << SignedVLC(uv[i]);
    ss << vlc::flush << vlc::align;
  } catch (const std::length_error &) {
    return -1;
  }
  const std::string s = ss.str();
  if ((long)s.size() > cap) return -2;
  std::memcpy(out, s.data(), s.size());
  return (long)s.size();
}

// ... and read back the way LDSliceIO reads it (Slices.cpp:246-303).  Returns bytes consumed.
long ref_ld_slice_read(const uint8_t *in, long len, int slice_bytes, int uv_split_bits, int ny, int nuv, int *q_index,
                       int *y_bits, int32_t *y, int32_t *uv) {
  std::istringstream ss(std::string(reinterpret_cast<const char *>(in), (size_t)len));
  Bits q(7);
  ss >> q;
  *q_index = (int)(unsigned)q;
  Bits yb((unsigned)uv_split_bits);
  ss >> yb;
  *y_bits = (int)(unsigned)yb;
  const int uv_bits = 8 * slice_bytes - 7 - uv_split_bits - *y_bits;
  ss >> vlc::bounded(*y_bits);
  SignedVLC v;
  for (int i = 0; i < ny; ++i) { ss >> v; y[i] = v; }
  ss >> vlc::flush;
  ss >> vlc::bounded(uv_bits);
  for (int i = 0; i < nuv; ++i) { ss >> v; uv[i] = v; }
  ss >> vlc::flush >> vlc::align;
  return (long)ss.tellg();
}

// Picture header + transform parameters of an HQ picture as HQWrappedPictureIO writes them (DataUnit.cpp:236-259):
// Bytes(4, picture number), unbounded, UnsignedVLC wavelet / depth, (major version >= 3: two false Booleans),
// UnsignedVLC slices_x / slices_y / prefix / scalar, Boolean(false), align.  Returns bytes written.
long ref_hq_picture_header(unsigned long picture_number, int major_version, unsigned wavelet, unsigned depth, unsigned slices_x,
                           unsigned slices_y, unsigned prefix, unsigned scalar, uint8_t *out, long cap) {
  std::ostringstream ss;
  ss << Bytes(4, picture_number);
  ss << vlc::unbounded << UnsignedVLC(wavelet) << UnsignedVLC(depth);
  if (major_version >= 3) ss << Boolean(false) << Boolean(false);
  ss << UnsignedVLC(slices_x) << UnsignedVLC(slices_y) << UnsignedVLC(prefix) << UnsignedVLC(scalar) << Boolean(false) << vlc::align;
  const std::string s = ss.str();
  if ((long)s.size() > cap) return -2;
  std::memcpy(out, s.data(), s.size());
  return (long)s.size();
}

// A run of UnsignedVLC values and Booleans (kind[i] 0: UnsignedVLC(value[i]), 1: Boolean(value[i] != 0)), then align:
// what the sequence header writer is made of (DataUnit.cpp:785-1000)
long ref_uvlc_bool_string(const uint32_t *value, const uint8_t *kind, int n, uint8_t *out, long cap) {
  std::ostringstream ss;
  ss << vlc::unbounded;
  for (int i = 0; i < n; ++i) {
    if (kind[i]) ss << Boolean(value[i] != 0);
    else ss << UnsignedVLC(value[i]);
  }
  ss << vlc::align;
  const std::string s = ss.str();
  if ((long)s.size() > cap) return -2;
  std::memcpy(out, s.data(), s.size());
  return (long)s.size();
}

int ref_svlc_numbits(int32_t value) { return (int)SignedVLC(value).numOfBits(); }
unsigned ref_svlc_code(int32_t value) { return SignedVLC(value).code(); }
int ref_uvlc_numbits(uint32_t value) { return (int)UnsignedVLC(value).numOfBits(); }
unsigned ref_uvlc_code(uint32_t value) { return UnsignedVLC(value).code(); }
}
