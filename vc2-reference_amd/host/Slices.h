// Slices.h -- slice byte budgeting and slice (de)serialisation of whole pictures
// (/root/reference/src/Library/Slices.h:17-127, src/Slices.cpp:28-49, :645-694) on libvc2hip.
#ifndef VC2HOST_SLICES_H
#define VC2HOST_SLICES_H
#include <iosfwd>
#include <vector>

#include "Arrays.h"
#include "Picture.h"

// slice_bytes(ySlices, xSlices, totalBytes, scalar), Slices.cpp:28-49
const Array2D slice_bytes(int ySlices, int xSlices, int totalBytes, int scalar);
// the size of the slice at (v, h) for a rational bytes-per-slice budget, Slices.cpp:18-26
int slice_bytes(int v, int h, int ySlices, int xSlices, int sliceBytesNumerator, int sliceBytesDenominator);

// bits of an LD slice's luma / chroma codes up to the last non-zero coefficient, Slices.cpp:51-95
int luma_slice_bits(const Array2D &lumaSlice, char waveletDepth);
int chroma_slice_bits(const Array2D &uSlice, const Array2D &vSlice, char waveletDepth);
// bytes of one HQ slice component in whole units of `scalar`, Slices.cpp:97-119 (throws "Slice scalar is too small, ...")
int component_slice_bytes(const Array2D &componentSlice, char waveletDepth, int scalar);

// ---- the reference's slice I/O surface (Slices.h:65-127): a picture's slices as stream-insertable objects.  The work is
// done by libvc2hip for the whole picture at once (vc2hip_hq_pack / _unpack, vc2hip_ld_pack / _unpack); these types only
// carry it through the iostream idiom the reference's tools are written in:
//   outStream << sliceio::highQualityVBR(prefix, scalar) << Slices(split_into_blocks(quantised, ys, xs), depth, qIndices);
struct Slices {
  Slices(const PictureArray &yuvSlices, int waveletDepth, const Array2D &qIndices)
      : yuvSlices(yuvSlices), waveletDepth(waveletDepth), qIndices(qIndices) {}
  Slices(const PictureFormat &pictureFormat, int waveletDepth, int ySlices, int xSlices);
  PictureArray yuvSlices;
  const int waveletDepth;
  Array2D qIndices;
  int nSlices() { return (int)(yuvSlices.shape()[0] * yuvSlices.shape()[1]); }
};
std::ostream &operator<<(std::ostream &stream, const Slices &s);
std::istream &operator>>(std::istream &stream, Slices &s);

struct Slice {
  Slice(const Picture &p, int d, int i) : yuvSlice(p), waveletDepth(d), qIndex(i) {}
  Slice(const PictureFormat &f, int d) : yuvSlice(f), waveletDepth(d), qIndex(0) {}
  Picture yuvSlice;
  const int waveletDepth;
  int qIndex;
};
std::ostream &operator<<(std::ostream &stream, const Slice &s); // one slice: the size comes from setBytes (CBR / LD)
std::istream &operator>>(std::istream &stream, Slice &s);

namespace sliceio {
enum SliceIOMode { UNKNOWN, LD, HQVBR, HQCBR };
SliceIOMode &sliceIOMode(std::ios_base &stream);
class lowDelay {
 public:
  lowDelay(const Array2D &b) : bytes(b) {}
  void operator()(std::ios_base &stream) const;
 private:
  const Array2D &bytes;
};
class highQualityCBR {
 public:
  highQualityCBR(const Array2D &b, int p, int s) : bytes(b), prefix(p), scalar(s) {}
  void operator()(std::ios_base &stream) const;
 private:
  const Array2D &bytes;
  const int prefix, scalar;
};
class highQualityVBR {
 public:
  highQualityVBR(int p, int s) : prefix(p), scalar(s) {}
  void operator()(std::ios_base &stream) const;
 private:
  const int prefix, scalar;
};
class setBytes {
 public:
  setBytes(int b) : bytes(b) {}
  void operator()(std::ios_base &stream) const;
 private:
  const int bytes;
};
const Array2D *SliceSizes(std::ios_base &stream);
} // namespace sliceio
std::ostream &operator<<(std::ostream &stream, sliceio::setBytes arg);
std::istream &operator>>(std::istream &stream, sliceio::setBytes arg);
std::ostream &operator<<(std::ostream &stream, sliceio::lowDelay arg);
std::istream &operator>>(std::istream &stream, sliceio::lowDelay arg);
std::ostream &operator<<(std::ostream &stream, sliceio::highQualityCBR arg);
std::istream &operator>>(std::istream &stream, sliceio::highQualityCBR arg);
std::ostream &operator<<(std::ostream &stream, sliceio::highQualityVBR arg);
std::istream &operator>>(std::istream &stream, sliceio::highQualityVBR arg);

// quantIndicesCBR(coefficients, qMatrix, sliceBytes, scalar), EncodeStream.cpp:73-125
const Array2D quantIndicesCBR(const Picture &coefficients, const Array1D &qMatrix, const Array2D &sliceBytes, int scalar);

// quantIndicesLD(coefficients, qMatrix, sliceBytes), EncodeStream.cpp:141-245
const Array2D quantIndicesLD(const Picture &coefficients, const Array1D &qMatrix, const Array2D &sliceBytes);
// operator<<(ostream, Slices) under lowDelay(sliceBytes), Slices.cpp:645-660 over :195-244
std::vector<unsigned char> packSlicesLD(const Picture &quantised, int waveletDepth, const Array2D &qIndices,
                                        const Array2D &sliceBytes);
// operator<<(ostream, Slices) under highQualityVBR / highQualityCBR: bytes of all slices of a picture.
// sliceBytes == nullptr selects VBR.
std::vector<unsigned char> packSlicesHQ(const Picture &quantised, int waveletDepth, const Array2D &qIndices,
                                        int prefix, int scalar, const Array2D *sliceBytes);
// operator>>(istream, Slices) under highQualityVBR / lowDelay: fills the quantised planes + indices
void unpackSlicesHQ(const unsigned char *data, std::size_t len, Picture &quantised, int waveletDepth,
                    Array2D &qIndices, int prefix, int scalar, std::size_t *consumed);
void unpackSlicesLD(const unsigned char *data, std::size_t len, Picture &quantised, int waveletDepth,
                    Array2D &qIndices, const Array2D &sliceBytes, std::size_t *consumed);
#endif
