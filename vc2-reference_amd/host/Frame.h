// Frame.h -- field access of interlaced frames (/root/reference/src/Library/Frame.h,
// src/Frame.cpp:40-110): the top field is the even rows of every plane, the bottom field the odd rows.
// Both the sample planes (diagnostic outputs) and the raw planar words (fused device path) are served.
#ifndef VC2HOST_FRAME_H
#define VC2HOST_FRAME_H
#include "Picture.h"

// The reference's Frame (Frame.h, Frame.cpp:40-110): a Picture whose fields are views of alternate rows.  The bodies
// in Frame.cpp are written with the reference's own expressions (y()[indices[Range(top, bottom, 2)][Range()]]).
class Frame : public Picture {
 public:
  Frame(const PictureFormat &f, bool interlaced = false, bool topFieldFirst = true) : Picture(f), ilaced(interlaced), tff(topFieldFirst) {}
  bool interlaced() const { return ilaced; }
  void interlaced(bool i) { ilaced = i; }
  bool topFieldFirst() const { return tff; }
  void topFieldFirst(bool t) { tff = t; }
  const Picture topField() const;
  void topField(const Picture &f);
  const Picture bottomField() const;
  void bottomField(const Picture &f);
  const Picture firstField() const { return tff ? topField() : bottomField(); }
  void firstField(const Picture &f) { if (tff) topField(f); else bottomField(f); }
  const Picture secondField() const { return tff ? bottomField() : topField(); }
  void secondField(const Picture &f) { if (tff) bottomField(f); else topField(f); }
  const Frame &frame() const { return *this; }
  void frame(const Picture &p) { y(p.y()); c1(p.c1()); c2(p.c2()); }

 private:
  bool ilaced, tff;
};

// format of one field of a frame (Frame.cpp:41-44: half the luma height; chroma follows the colour format)
PictureFormat fieldFormat(const PictureFormat &frame);
const Picture fieldOf(const Picture &frame, bool top);          // Frame::topField()/bottomField() const
void setField(Picture &frame, const Picture &field, bool top);   // Frame::topField(f)/bottomField(f)
// the same on raw planar words, wordBytes per sample, planes Y,U,V back to back
void extractFieldRaw(const unsigned char *frame, const PictureFormat &frameFormat, int wordBytes, bool top,
                     unsigned char *field);
void insertFieldRaw(unsigned char *frame, const PictureFormat &frameFormat, int wordBytes, bool top,
                    const unsigned char *field);
#endif
