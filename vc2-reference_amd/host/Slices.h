// Slices.h -- slice byte budgeting and slice (de)serialisation of whole pictures
// (/root/reference/src/Library/Slices.h:17-127, src/Slices.cpp:28-49, :645-694) on libvc2hip.
#ifndef VC2HOST_SLICES_H
#define VC2HOST_SLICES_H
#include <vector>

#include "Arrays.h"
#include "Picture.h"

// slice_bytes(ySlices, xSlices, totalBytes, scalar), Slices.cpp:28-49
const Array2D slice_bytes(int ySlices, int xSlices, int totalBytes, int scalar);

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
