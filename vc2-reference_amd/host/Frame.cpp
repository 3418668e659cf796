#include "Frame.h"

#include <cstring>
#include <stdexcept>

PictureFormat fieldFormat(const PictureFormat &f) {
  if ((f.lumaHeight() & 1) || (f.chromaHeight() & 1))
    throw std::logic_error("interlaced coding needs even luma and chroma heights");
  return PictureFormat(f.lumaHeight() / 2, f.lumaWidth(), f.chromaFormat());
}

// ---- Frame: the reference's bodies (Frame.cpp:40-94), expression for expression, on host/Arrays.h
const Picture Frame::topField() const {
  // Get parameters of field
  const int height = format().lumaHeight() / 2;
  const int width = format().lumaWidth();
  const ColourFormat chromaFormat = format().chromaFormat();
  // Construct interlaced field
  Picture picture(PictureFormat(height, width, chromaFormat));
  // Set components of interlaced field
  const int top = 0;
  const int yBottom = format().lumaHeight();
  const int uvBottom = format().chromaHeight();
  picture.y(y()[indices[Range(top, yBottom, 2)][Range()]]);
  picture.c1(c1()[indices[Range(top, uvBottom, 2)][Range()]]);
  picture.c2(c2()[indices[Range(top, uvBottom, 2)][Range()]]);
  return picture;
}

void Frame::topField(const Picture &f) {
  const int top = 0;
  const int yBottom = format().lumaHeight();
  const int uvBottom = format().chromaHeight();
  luma[indices[Range(top, yBottom, 2)][Range()]] = f.y();
  chroma1[indices[Range(top, uvBottom, 2)][Range()]] = f.c1();
  chroma2[indices[Range(top, uvBottom, 2)][Range()]] = f.c2();
}

const Picture Frame::bottomField() const {
  const int height = format().lumaHeight() / 2;
  const int width = format().lumaWidth();
  const ColourFormat chromaFormat = format().chromaFormat();
  Picture picture(PictureFormat(height, width, chromaFormat));
  const int top = 1;
  const int yBottom = format().lumaHeight();
  const int uvBottom = format().chromaHeight();
  picture.y(y()[indices[Range(top, yBottom, 2)][Range()]]);
  picture.c1(c1()[indices[Range(top, uvBottom, 2)][Range()]]);
  picture.c2(c2()[indices[Range(top, uvBottom, 2)][Range()]]);
  return picture;
}

void Frame::bottomField(const Picture &f) {
  const int top = 1;
  const int yBottom = format().lumaHeight();
  const int uvBottom = format().chromaHeight();
  luma[indices[Range(top, yBottom, 2)][Range()]] = f.y();
  chroma1[indices[Range(top, uvBottom, 2)][Range()]] = f.c1();
  chroma2[indices[Range(top, uvBottom, 2)][Range()]] = f.c2();
}

static Array2D rowsOf(const Array2D &a, int first) {
  const Index h = a.shape()[0], w = a.shape()[1];
  Array2D out((h - first + 1) / 2, w);
  for (Index y = first, r = 0; y < h; y += 2, ++r) std::memcpy(out[r], a[y], (std::size_t)w * sizeof(int));
  return out;
}
static void setRows(Array2D &a, const Array2D &f, int first) {
  const Index h = a.shape()[0], w = a.shape()[1];
  for (Index y = first, r = 0; y < h && r < f.shape()[0]; y += 2, ++r) std::memcpy(a[y], f[r], (std::size_t)w * sizeof(int));
}

const Picture fieldOf(const Picture &frame, bool top) {
  Picture p(fieldFormat(frame.format()));
  const int first = top ? 0 : 1;
  p.y(rowsOf(frame.y(), first)); p.c1(rowsOf(frame.c1(), first)); p.c2(rowsOf(frame.c2(), first));
  return p;
}
void setField(Picture &frame, const Picture &field, bool top) {
  const int first = top ? 0 : 1;
  Array2D y(frame.y()), u(frame.c1()), v(frame.c2());
  setRows(y, field.y(), first); setRows(u, field.c1(), first); setRows(v, field.c2(), first);
  frame.y(y); frame.c1(u); frame.c2(v);
}

template <bool EXTRACT>
static void fieldRaw(unsigned char *frame, const PictureFormat &f, int wordBytes, bool top, unsigned char *field) {
  const int first = top ? 0 : 1;
  const int hs[3] = {f.lumaHeight(), f.chromaHeight(), f.chromaHeight()};
  const int ws[3] = {f.lumaWidth(), f.chromaWidth(), f.chromaWidth()};
  for (int c = 0; c < 3; ++c) {
    const std::size_t row = (std::size_t)ws[c] * wordBytes;
    for (int y = first; y < hs[c]; y += 2, field += row) {
      if (EXTRACT) std::memcpy(field, frame + (std::size_t)y * row, row);
      else std::memcpy(frame + (std::size_t)y * row, field, row);
    }
    frame += (std::size_t)hs[c] * row;
  }
}
void extractFieldRaw(const unsigned char *frame, const PictureFormat &f, int wordBytes, bool top, unsigned char *field) {
  fieldRaw<true>(const_cast<unsigned char *>(frame), f, wordBytes, top, field);
}
void insertFieldRaw(unsigned char *frame, const PictureFormat &f, int wordBytes, bool top, const unsigned char *field) {
  fieldRaw<false>(frame, f, wordBytes, top, const_cast<unsigned char *>(field));
}
