#include "Picture.h"

#include <ostream>
#include <stdexcept>

std::ostream &operator<<(std::ostream &os, ColourFormat format) {
  switch (format) {
    case CF444: return os << "4:4:4";
    case CF422: return os << "4:2:2";
    case CF420: return os << "4:2:0";
    default: return os << "Unknown colour format!";
  }
}
ColourFormat parseColourFormat(const std::string &text) {
  if (text == "4:4:4") return CF444;
  if (text == "4:2:2") return CF422;
  if (text == "4:2:0") return CF420;
  throw std::invalid_argument("invalid colour format");
}

void PictureFormat::construct(int height, int width, ColourFormat cf) {
  yHeight = height;
  yWidth = width;
  uvFormat = cf;
  switch (cf) {
    case CF444: uvHeight = height; uvWidth = width; break;
    case CF422: uvHeight = height; uvWidth = width / 2; break;
    case CF420: uvHeight = height / 2; uvWidth = width / 2; break;
    case CF_UNSET: uvHeight = 0; uvWidth = 0; break;
    default: throw std::invalid_argument("Invalid colour format");
  }
}

const Picture clip(const Picture &p, int yMin, int yMax, int uvMin, int uvMax) {
  Picture out(p.format());
  out.y(clip(p.y(), yMin, yMax));
  out.c1(clip(p.c1(), uvMin, uvMax));
  out.c2(clip(p.c2(), uvMin, uvMax));
  return out;
}

void unpackSamples(const unsigned char *raw, int wordBytes, int bitDepth, bool leftJustified, bool offsetBinary,
                   Array2D &plane) {
  const int shift = leftJustified ? 8 * wordBytes - bitDepth : 0;
  const int offset = offsetBinary ? 1 << (bitDepth - 1) : 0;
  int *d = plane.data();
  for (std::size_t i = 0; i < plane.num_elements(); ++i) {
    unsigned v = 0;
    for (int b = 0; b < wordBytes; ++b) v = (v << 8) | raw[i * wordBytes + b];
    if (!offsetBinary && wordBytes < 4) { // two's complement words: sign-extend, arithmetic shift
      int sv = (int)(v << (32 - 8 * wordBytes)) >> (32 - 8 * wordBytes);
      d[i] = sv >> shift;
    } else {
      d[i] = (int)(v >> shift) - offset;
    }
  }
}
void packSamples(const Array2D &plane, int wordBytes, int bitDepth, bool leftJustified, bool offsetBinary,
                 unsigned char *raw) {
  const int shift = leftJustified ? 8 * wordBytes - bitDepth : 0;
  const int offset = offsetBinary ? 1 << (bitDepth - 1) : 0;
  const int *s = plane.data();
  for (std::size_t i = 0; i < plane.num_elements(); ++i) {
    const unsigned v = (unsigned)(s[i] + offset) << shift;
    for (int b = 0; b < wordBytes; ++b) raw[i * wordBytes + b] = (unsigned char)(v >> (8 * (wordBytes - 1 - b)));
  }
}
