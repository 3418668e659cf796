#include "Frame.h"

#include <cstring>
#include <stdexcept>

PictureFormat fieldFormat(const PictureFormat &f) {
  if ((f.lumaHeight() & 1) || (f.chromaHeight() & 1))
    throw std::logic_error("interlaced coding needs even luma and chroma heights");
  return PictureFormat(f.lumaHeight() / 2, f.lumaWidth(), f.chromaFormat());
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

// ---- class Frame (surface: Frame.h:18-40).  A field is every second row of each plane; all four accessors are the two
// row-copy helpers above with the parity of the first row chosen by the caller.
Frame::Frame(int height, int width, ColourFormat cf, bool interlaced, bool topFieldFirst)
    : Picture(PictureFormat(height, width, cf)), intl(interlaced), tff(topFieldFirst) {}
Frame::Frame(const PictureFormat &f, bool interlaced, bool topFieldFirst) : Picture(f), intl(interlaced), tff(topFieldFirst) {}
bool Frame::interlaced() const { return intl; }
void Frame::interlaced(bool on) { intl = on; }
bool Frame::topFieldFirst() const { return tff; }
void Frame::topFieldFirst(bool on) { tff = on; }

const Picture Frame::topField() const { return fieldOf(*this, true); }
const Picture Frame::bottomField() const { return fieldOf(*this, false); }
void Frame::topField(const Picture &field) {
  setRows(luma, field.y(), 0); setRows(chroma1, field.c1(), 0); setRows(chroma2, field.c2(), 0);
}
void Frame::bottomField(const Picture &field) {
  setRows(luma, field.y(), 1); setRows(chroma1, field.c1(), 1); setRows(chroma2, field.c2(), 1);
}
const Picture Frame::firstField() const { return fieldOf(*this, tff); }
const Picture Frame::secondField() const { return fieldOf(*this, !tff); }
void Frame::firstField(const Picture &field) { if (tff) topField(field); else bottomField(field); }
void Frame::secondField(const Picture &field) { if (tff) bottomField(field); else topField(field); }
const Picture &Frame::frame() const { return *this; }
void Frame::frame(const Picture &whole) { Picture::operator=(whole); }
