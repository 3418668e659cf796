// WaveletTransform.h -- same free functions as /root/reference/src/Library/WaveletTransform.h:26-77,
// implemented on libvc2hip (no CPU path).
#ifndef VC2HOST_WAVELETTRANSFORM_H
#define VC2HOST_WAVELETTRANSFORM_H

#include <iosfwd>
#include <string>

#include "Arrays.h"
#include "Picture.h"

enum WaveletKernel { DD97, LeGall, DD137, Haar0, Haar1, Fidelity, Daub97, NullKernel };
std::ostream &operator<<(std::ostream &os, WaveletKernel kernel);
WaveletKernel parseWaveletKernel(const std::string &text); // throws "invalid wavelet kernel"

int paddedSize(int size, int depth);                                                    // :74-77
int sliceSizeIsValid(int waveletDepth, int lengthLuma, int lengthChroma, int nSize);    // :116-136
const Array1D quantMatrix(WaveletKernel kernel, int depth);                                   // :345-423

const BlockVector split_into_subbands(const Array2D &picture, char waveletDepth);             // :428-450
const Array2D merge_subbands(const BlockVector &subbands);                                   // :454-476

const Array2D waveletTransform(const Array2D &picture, WaveletKernel kernel, int depth);      // :262-281
const Array2D inverseWaveletTransform(const Array2D &transform, WaveletKernel kernel, int depth, Shape2D shape); // :321-342
const Picture waveletTransform(const Picture &picture, WaveletKernel kernel, int depth);      // :1267-1279
const Picture inverseWaveletTransform(const Picture &transform, WaveletKernel kernel, int depth,
                                      PictureFormat format);                                  // :1281-1292
#endif
