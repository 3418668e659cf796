// Frame.h -- field access of interlaced frames (/root/reference/src/Library/Frame.h,
// src/Frame.cpp:40-110): the top field is the even rows of every plane, the bottom field the odd rows.
// Both the sample planes (diagnostic outputs) and the raw planar words (fused device path) are served.
#ifndef VC2HOST_FRAME_H
#define VC2HOST_FRAME_H
#include "Picture.h"

// Frame: a Picture that also knows whether it is interlaced and which field comes first.  The class surface is the
// reference's (Frame.h:18-40: same member functions, out of line, same data member names), so that a translation unit
// written against the reference's header -- the reference's own src/Frame.cpp included -- builds against this one
// (tests/test_host.py compiles it from /root/reference when that checkout exists).  The bodies in host/Frame.cpp are
// ours: whole-row copies through fieldOf / setField below.
class Frame : public Picture {
 public:
  Frame(int height, int width, ColourFormat, bool interlaced = false, bool topFieldFirst = true);
  Frame(const PictureFormat &, bool interlaced = false, bool topFieldFirst = true);
  bool interlaced() const;
  void interlaced(bool);
  bool topFieldFirst() const;
  void topFieldFirst(bool);
  const Picture topField() const;
  void topField(const Picture &);
  const Picture bottomField() const;
  void bottomField(const Picture &);
  const Picture firstField() const;
  void firstField(const Picture &);
  const Picture secondField() const;
  void secondField(const Picture &);
  const Picture &frame() const;
  void frame(const Picture &);

 private:
  bool intl;
  bool tff;
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
