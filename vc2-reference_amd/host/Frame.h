// Frame.h -- field access of interlaced frames (/root/reference/src/Library/Frame.h,
// src/Frame.cpp:40-110): the top field is the even rows of every plane, the bottom field the odd rows.
// Both the sample planes (diagnostic outputs) and the raw planar words (fused device path) are served.
#ifndef VC2HOST_FRAME_H
#define VC2HOST_FRAME_H
#include "Picture.h"

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
