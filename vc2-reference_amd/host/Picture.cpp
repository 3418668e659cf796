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

// Picture.cpp:231-247: slice (v, h) of every component
const PictureArray split_into_blocks(const Picture &picture, int ySlices, int xSlices) {
  const BlockArray y = split_into_blocks(picture.y(), ySlices, xSlices);
  const BlockArray u = split_into_blocks(picture.c1(), ySlices, xSlices);
  const BlockArray v = split_into_blocks(picture.c2(), ySlices, xSlices);
  const PictureFormat f((int)y.at(0, 0).shape()[0], (int)y.at(0, 0).shape()[1], (int)u.at(0, 0).shape()[0],
                        (int)u.at(0, 0).shape()[1], picture.format().chromaFormat());
  PictureArray out(ySlices, xSlices);
  for (int i = 0; i < ySlices; ++i)
    for (int j = 0; j < xSlices; ++j) {
      Picture p(f);
      p.y(y.at(i, j)); p.c1(u.at(i, j)); p.c2(v.at(i, j));
      out.at(i, j) = p;
    }
  return out;
}

// Picture.cpp:249-271
const Picture merge_blocks(const PictureArray &blocks) {
  const int ys = (int)blocks.shape()[0], xs = (int)blocks.shape()[1];
  BlockArray y(ys, xs), u(ys, xs), v(ys, xs);
  for (int i = 0; i < ys; ++i)
    for (int j = 0; j < xs; ++j) { y.at(i, j) = blocks.at(i, j).y(); u.at(i, j) = blocks.at(i, j).c1(); v.at(i, j) = blocks.at(i, j).c2(); }
  const Array2D my = merge_blocks(y), mu = merge_blocks(u), mv = merge_blocks(v);
  const PictureFormat f((int)my.shape()[0], (int)my.shape()[1], (int)mu.shape()[0], (int)mu.shape()[1],
                        blocks.at(0, 0).format().chromaFormat());
  Picture out(f);
  out.y(my); out.c1(mu); out.c2(mv);
  return out;
}
