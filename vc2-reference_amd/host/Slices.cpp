#include "Slices.h"

#include <istream>
#include <iterator>
#include <ostream>
#include <stdexcept>

#include "Hip.h"
#include "WaveletTransform.h"

const Array2D slice_bytes(int ySlices, int xSlices, int totalBytes, int scalar) {
  Array2D b(ySlices, xSlices);
  vc2hip_slice_bytes(ySlices, xSlices, totalBytes, scalar, b.data());
  return b;
}

static vc2hip_geom geomOf(const Picture &p, int depth, const Array2D &q) {
  vc2hip_geom g = {(int)p.y().shape()[0], (int)p.y().shape()[1], (int)p.c1().shape()[0], (int)p.c1().shape()[1], depth,
                   (int)q.shape()[0], (int)q.shape()[1]};
  return g;
}

const Array2D quantIndicesCBR(const Picture &t, const Array1D &qMatrix, const Array2D &sliceBytes, int scalar) {
  Array2D q(sliceBytes.shape());
  const vc2hip_geom g = geomOf(t, (int)(qMatrix.size() - 1) / 3, sliceBytes);
  vc2hip_ctx *c = hipContext();
  hipCheck(c, vc2hip_cbr_qindices(c, t.y().data(), t.c1().data(), t.c2().data(), &g, qMatrix.data(), sliceBytes.data(),
                                  scalar, q.data()));
  return q;
}

const Array2D quantIndicesLD(const Picture &t, const Array1D &qMatrix, const Array2D &sliceBytes) {
  Array2D q(sliceBytes.shape());
  const vc2hip_geom g = geomOf(t, (int)(qMatrix.size() - 1) / 3, sliceBytes);
  vc2hip_ctx *c = hipContext();
  hipCheck(c, vc2hip_ld_qindices(c, t.y().data(), t.c1().data(), t.c2().data(), &g, qMatrix.data(), sliceBytes.data(), q.data()));
  return q;
}

std::vector<unsigned char> packSlicesLD(const Picture &qp, int depth, const Array2D &qIndices, const Array2D &sliceBytes) {
  const vc2hip_geom g = geomOf(qp, depth, qIndices);
  std::size_t cap = 0;
  for (std::size_t i = 0; i < sliceBytes.num_elements(); ++i) cap += (std::size_t)sliceBytes.data()[i];
  std::vector<unsigned char> out(cap + 64);
  std::size_t len = 0;
  vc2hip_ctx *c = hipContext();
  hipCheck(c, vc2hip_ld_pack(c, qp.y().data(), qp.c1().data(), qp.c2().data(), &g, qIndices.data(), sliceBytes.data(),
                             out.data(), out.size(), &len));
  out.resize(len);
  return out;
}

std::vector<unsigned char> packSlicesHQ(const Picture &qp, int depth, const Array2D &qIndices, int prefix, int scalar,
                                        const Array2D *sliceBytes) {
  const vc2hip_geom g = geomOf(qp, depth, qIndices);
  const std::size_t n = qIndices.num_elements();
  std::size_t cap = n * ((std::size_t)prefix + 4 + 3 * 255 * (std::size_t)scalar);
  std::vector<unsigned char> out(cap + 64);
  std::size_t len = 0;
  vc2hip_ctx *c = hipContext();
  hipCheck(c, vc2hip_hq_pack(c, qp.y().data(), qp.c1().data(), qp.c2().data(), &g, qIndices.data(), prefix, scalar,
                             sliceBytes ? sliceBytes->data() : nullptr, out.data(), out.size(), &len));
  out.resize(len);
  return out;
}

void unpackSlicesHQ(const unsigned char *data, std::size_t len, Picture &qp, int depth, Array2D &qIndices, int prefix,
                    int scalar, std::size_t *consumed) {
  const vc2hip_geom g = geomOf(qp, depth, qIndices);
  Array2D y(qp.y().shape()), u(qp.c1().shape()), v(qp.c2().shape());
  vc2hip_ctx *c = hipContext();
  hipCheck(c, vc2hip_hq_unpack(c, data, len, &g, prefix, scalar, y.data(), u.data(), v.data(), qIndices.data(), consumed));
  qp.y(y); qp.c1(u); qp.c2(v);
}

void unpackSlicesLD(const unsigned char *data, std::size_t len, Picture &qp, int depth, Array2D &qIndices,
                    const Array2D &sliceBytes, std::size_t *consumed) {
  const vc2hip_geom g = geomOf(qp, depth, qIndices);
  Array2D y(qp.y().shape()), u(qp.c1().shape()), v(qp.c2().shape());
  vc2hip_ctx *c = hipContext();
  hipCheck(c, vc2hip_ld_unpack(c, data, len, &g, sliceBytes.data(), y.data(), u.data(), v.data(), qIndices.data(), consumed));
  qp.y(y); qp.c1(u); qp.c2(v);
}

// ---- stream state of the manipulators (Slices.cpp:121-193 keeps it in iword / pword slots too) ------------------------
namespace {
const int kMode = std::ios_base::xalloc(), kPrefix = std::ios_base::xalloc(), kScalar = std::ios_base::xalloc(),
          kBytes = std::ios_base::xalloc(), kSingle = std::ios_base::xalloc();
}
namespace sliceio {
SliceIOMode &sliceIOMode(std::ios_base &stream) { return reinterpret_cast<SliceIOMode &>(stream.iword(kMode)); }
const Array2D *SliceSizes(std::ios_base &stream) { return static_cast<const Array2D *>(stream.pword(kBytes)); }
void lowDelay::operator()(std::ios_base &stream) const { stream.iword(kMode) = LD; stream.pword(kBytes) = const_cast<Array2D *>(&bytes); }
void highQualityCBR::operator()(std::ios_base &stream) const {
  stream.iword(kMode) = HQCBR; stream.pword(kBytes) = const_cast<Array2D *>(&bytes);
  stream.iword(kPrefix) = prefix; stream.iword(kScalar) = scalar;
}
void highQualityVBR::operator()(std::ios_base &stream) const {
  stream.iword(kMode) = HQVBR; stream.pword(kBytes) = nullptr;
  stream.iword(kPrefix) = prefix; stream.iword(kScalar) = scalar;
}
void setBytes::operator()(std::ios_base &stream) const { stream.iword(kSingle) = bytes; }
} // namespace sliceio
#define VC2_MANIP(T)                                                                          \
  std::ostream &operator<<(std::ostream &stream, sliceio::T arg) { arg(stream); return stream; } \
  std::istream &operator>>(std::istream &stream, sliceio::T arg) { arg(stream); return stream; }
VC2_MANIP(setBytes)
VC2_MANIP(lowDelay)
VC2_MANIP(highQualityCBR)
VC2_MANIP(highQualityVBR)
#undef VC2_MANIP

Slices::Slices(const PictureFormat &pictureFormat, int depth, int ySlices, int xSlices)
    : yuvSlices(ySlices, xSlices), waveletDepth(depth), qIndices(ySlices, xSlices) {
  const PictureFormat f(pictureFormat.lumaHeight() / ySlices, pictureFormat.lumaWidth() / xSlices,
                        pictureFormat.chromaHeight() / ySlices, pictureFormat.chromaWidth() / xSlices, pictureFormat.chromaFormat());
  for (int v = 0; v < ySlices; ++v)
    for (int h = 0; h < xSlices; ++h) yuvSlices.at(v, h) = Picture(f);
}

// operator<<(ostream, Slices), Slices.cpp:645-660: every slice of the picture in raster order, in the stream's mode
std::ostream &operator<<(std::ostream &stream, const Slices &s) {
  const Picture quantised = merge_blocks(s.yuvSlices);
  std::vector<unsigned char> bytes;
  switch (sliceio::sliceIOMode(stream)) {
    case sliceio::HQVBR:
      bytes = packSlicesHQ(quantised, s.waveletDepth, s.qIndices, (int)stream.iword(kPrefix), (int)stream.iword(kScalar), nullptr);
      break;
    case sliceio::HQCBR:
      bytes = packSlicesHQ(quantised, s.waveletDepth, s.qIndices, (int)stream.iword(kPrefix), (int)stream.iword(kScalar),
                           sliceio::SliceSizes(stream));
      break;
    case sliceio::LD:
      bytes = packSlicesLD(quantised, s.waveletDepth, s.qIndices, *sliceio::SliceSizes(stream));
      break;
    default:
      throw std::invalid_argument("invalid slice IO mode");
  }
  stream.write(reinterpret_cast<const char *>(bytes.data()), (std::streamsize)bytes.size());
  return stream;
}

// operator>>(istream, Slices), Slices.cpp:662-694 (whole pictures; fragments are reassembled by the tools before)
std::istream &operator>>(std::istream &stream, Slices &s) {
  const std::streampos start = stream.tellg();
  const std::vector<unsigned char> data((std::istreambuf_iterator<char>(stream)), std::istreambuf_iterator<char>());
  Picture quantised = merge_blocks(s.yuvSlices);
  std::size_t used = 0;
  switch (sliceio::sliceIOMode(stream)) {
    case sliceio::HQVBR:
    case sliceio::HQCBR: // the reference reads CBR pictures through the in-stream length bytes as well (DecodeStream.cpp:512)
      unpackSlicesHQ(data.data(), data.size(), quantised, s.waveletDepth, s.qIndices, (int)stream.iword(kPrefix),
                     (int)stream.iword(kScalar), &used);
      break;
    case sliceio::LD:
      unpackSlicesLD(data.data(), data.size(), quantised, s.waveletDepth, s.qIndices, *sliceio::SliceSizes(stream), &used);
      break;
    default:
      throw std::invalid_argument("invalid slice IO mode");
  }
  s.yuvSlices = split_into_blocks(quantised, (int)s.qIndices.shape()[0], (int)s.qIndices.shape()[1]);
  stream.clear();
  stream.seekg(start + (std::streamoff)used);
  return stream;
}

// a single slice: a 1 x 1 picture of slices (sizes from setBytes in the CBR / LD modes)
std::ostream &operator<<(std::ostream &stream, const Slice &s) {
  PictureArray one(1, 1);
  one.at(0, 0) = s.yuvSlice;
  Array2D q(1, 1), b(1, 1);
  q[0][0] = s.qIndex;
  b[0][0] = (int)stream.iword(kSingle);
  const void *saved = stream.pword(kBytes);
  if (sliceio::sliceIOMode(stream) != sliceio::HQVBR) stream.pword(kBytes) = &b;
  stream << Slices(one, s.waveletDepth, q);
  stream.pword(kBytes) = const_cast<void *>(saved);
  return stream;
}
std::istream &operator>>(std::istream &stream, Slice &s) {
  Slices one(s.yuvSlice.format(), s.waveletDepth, 1, 1);
  Array2D b(1, 1);
  b[0][0] = (int)stream.iword(kSingle);
  const void *saved = stream.pword(kBytes);
  if (sliceio::sliceIOMode(stream) != sliceio::HQVBR) stream.pword(kBytes) = &b;
  stream >> one;
  stream.pword(kBytes) = const_cast<void *>(saved);
  s.yuvSlice = one.yuvSlices.at(0, 0);
  s.qIndex = one.qIndices[0][0];
  return stream;
}
