// Picture.h -- PictureFormat / Picture with the reference's accessors
// (/root/reference/src/Library/Picture.h:17-110, src/Picture.cpp:49-73, :152-177, :231-292).
#ifndef VC2HOST_PICTURE_H
#define VC2HOST_PICTURE_H

#include <iosfwd>
#include <string>
#include <vector>

#include "Arrays.h"

enum ColourFormat { CF_UNSET = -1, CF444, CF422, CF420 };
std::ostream &operator<<(std::ostream &os, ColourFormat format);
ColourFormat parseColourFormat(const std::string &text); // "4:4:4" | "4:2:2" | "4:2:0", else throws

class PictureFormat {
 public:
  PictureFormat() { construct(0, 0, CF_UNSET); }
  PictureFormat(int height, int width, ColourFormat cf) { construct(height, width, cf); }
  PictureFormat(int lumaHeight, int lumaWidth, int chromaHeight, int chromaWidth, ColourFormat cf)
      : yHeight(lumaHeight), yWidth(lumaWidth), uvHeight(chromaHeight), uvWidth(chromaWidth), uvFormat(cf) {}
  int lumaHeight() const { return yHeight; }
  int lumaWidth() const { return yWidth; }
  int chromaHeight() const { return uvHeight; }
  int chromaWidth() const { return uvWidth; }
  ColourFormat chromaFormat() const { return uvFormat; }
  Shape2D lumaShape() const { Shape2D s = {{yHeight, yWidth}}; return s; }
  Shape2D chromaShape() const { Shape2D s = {{uvHeight, uvWidth}}; return s; }
  // samples of one picture
  long samples() const { return (long)yHeight * yWidth + 2L * uvHeight * uvWidth; }

 private:
  void construct(int height, int width, ColourFormat cf);
  int yHeight, yWidth, uvHeight, uvWidth;
  ColourFormat uvFormat;
};

class Picture {
 public:
  Picture() {}
  explicit Picture(const PictureFormat &f) : fmt(f), luma(f.lumaShape()), chroma1(f.chromaShape()), chroma2(f.chromaShape()) {}
  const PictureFormat &format() const { return fmt; }
  const Array2D &y() const { return luma; }
  const Array2D &c1() const { return chroma1; }
  const Array2D &c2() const { return chroma2; }
  void y(const Array2D &a) { luma = a; }
  void c1(const Array2D &a) { chroma1 = a; }
  void c2(const Array2D &a) { chroma2 = a; }

 protected: // (Frame writes its fields into them, Frame.cpp:65-74)
  PictureFormat fmt;
  Array2D luma, chroma1, chroma2;
};

// ySlices x xSlices pictures (the reference's PictureArray, a 2-D multi_array of Picture)
class PictureArray {
 public:
  PictureArray(int y = 0, int x = 0) : ys_(y), xs_(x), p_((std::size_t)(y * x)) { d_[0] = y; d_[1] = x; }
  const Index *shape() const { return dims(); }
  Picture &at(int v, int h) { return p_[(std::size_t)(v * xs_ + h)]; }
  const Picture &at(int v, int h) const { return p_[(std::size_t)(v * xs_ + h)]; }
  Picture *operator[](int v) { return p_.data() + (std::size_t)v * xs_; }
  const Picture *operator[](int v) const { return p_.data() + (std::size_t)v * xs_; }

 private:
  const Index *dims() const { d_[0] = ys_; d_[1] = xs_; return d_; }
  int ys_, xs_;
  mutable Index d_[2];
  std::vector<Picture> p_;
};

const PictureArray split_into_blocks(const Picture &picture, int ySlices, int xSlices); // Picture.cpp:231-247
const Picture merge_blocks(const PictureArray &blocks);                                // Picture.cpp:249-271

const Picture clip(const Picture &p, int yMin, int yMax, int uvMin, int uvMax); // Picture.cpp:284-292

// raw planar words <-> planes (Arrays.cpp:333-426 under the manipulators the tools set)
void unpackSamples(const unsigned char *raw, int wordBytes, int bitDepth, bool leftJustified, bool offsetBinary,
                   Array2D &plane);
void packSamples(const Array2D &plane, int wordBytes, int bitDepth, bool leftJustified, bool offsetBinary,
                 unsigned char *raw);
#endif
