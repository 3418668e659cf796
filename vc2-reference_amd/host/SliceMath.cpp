// SliceMath.cpp -- the arithmetic of the reference's slice surface that involves no picture-sized work: sizes and bit
// counts of single slices (/root/reference/src/Library/src/Slices.cpp:18-26, :51-119).
#include <stdexcept>

#include "Slices.h"
#include "WaveletTransform.h"

// ---- host-side arithmetic of the slice surface (no picture data moves here) -------------------------------------------
// Slices.cpp:18-26
int slice_bytes(int v, int h, int /*ySlices*/, int xSlices, int sliceBytesNumerator, int sliceBytesDenominator) {
  const long sliceNumber = (long)v * xSlices + h;
  long bytes = ((sliceNumber + 1) * sliceBytesNumerator) / sliceBytesDenominator;
  bytes -= (sliceNumber * sliceBytesNumerator) / sliceBytesDenominator;
  return (int)bytes;
}

static int svlcBits(int v) { // SignedVLC(v).numOfBits(), VLC.cpp:78-85
  if (v == 0) return 1;
  unsigned m = (v < 0 ? 0u - (unsigned)v : (unsigned)v) + 1u;
  int k = 0;
  while (m >>= 1) ++k;
  return 2 * k + 2;
}
static int bitsToLastNonZero(const BlockVector &a, const BlockVector *b) {
  int count = 0, gross = 0;
  for (std::size_t band = 0; band < a.size(); ++band)
    for (Index y = 0; y < a[band].shape()[0]; ++y)
      for (Index x = 0; x < a[band].shape()[1]; ++x) {
        int n = svlcBits(a[band][y][x]);
        gross += n;
        if (n > 1) count = gross;
        if (b) {
          n = svlcBits((*b)[band][y][x]);
          gross += n;
          if (n > 1) count = gross;
        }
      }
  return count;
}
int luma_slice_bits(const Array2D &lumaSlice, char waveletDepth) {
  return bitsToLastNonZero(split_into_subbands(lumaSlice, waveletDepth), nullptr);
}
int chroma_slice_bits(const Array2D &uSlice, const Array2D &vSlice, char waveletDepth) {
  const BlockVector v = split_into_subbands(vSlice, waveletDepth);
  return bitsToLastNonZero(split_into_subbands(uSlice, waveletDepth), &v);
}
int component_slice_bytes(const Array2D &componentSlice, char waveletDepth, int scalar) {
  const int count = bitsToLastNonZero(split_into_subbands(componentSlice, waveletDepth), nullptr);
  const int scaled = ((count + 7) / 8 + scalar - 1) / scalar;
  if (scaled > 0xFF) throw std::logic_error("Slice scalar is too small, consider using a larger slice scalar.");
  return scaled * scalar;
}

