// Quantisation.h -- /root/reference/src/Library/Quantisation.h:94-114 on libvc2hip.
#ifndef VC2HOST_QUANTISATION_H
#define VC2HOST_QUANTISATION_H
#include "Arrays.h"
#include "Picture.h"

// LD: DC-predicted LL band (Quantisation.h:86-88, Quantisation.cpp:358-367)
const Array2D quantise_transform(const Array2D &coefficients, const Array2D &qIndices, const Array1D &qMatrix);
const Picture quantise_transform(const Picture &transform, const Array2D &qIndices, const Array1D &qMatrix);
const Array2D quantise_transform_np(const Array2D &coefficients, const Array2D &qIndices, const Array1D &qMatrix);
const Array2D inverse_quantise_transform_np(const Array2D &qCoeffs, const Array2D &qIndices, const Array1D &qMatrix);
const Array2D inverse_quantise_transform(const Array2D &qCoeffs, const Array2D &qIndices, const Array1D &qMatrix); // LD
const Picture quantise_transform_np(const Picture &transform, const Array2D &qIndices, const Array1D &qMatrix);
const Picture inverse_quantise_transform_np(const Picture &qCoeffs, const Array2D &qIndices, const Array1D &qMatrix);
const Picture inverse_quantise_transform(const Picture &qCoeffs, const Array2D &qIndices, const Array1D &qMatrix);
#endif
