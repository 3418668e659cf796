// Quantisation.h -- /root/reference/src/Library/Quantisation.h:94-114 on libvc2hip.
#ifndef VC2HOST_QUANTISATION_H
#define VC2HOST_QUANTISATION_H
#include "Arrays.h"
#include "Picture.h"

const Array2D quantise_transform_np(const Array2D &coefficients, const Array2D &qIndices, const Array1D &qMatrix);
const Array2D inverse_quantise_transform_np(const Array2D &qCoeffs, const Array2D &qIndices, const Array1D &qMatrix);
const Array2D inverse_quantise_transform(const Array2D &qCoeffs, const Array2D &qIndices, const Array1D &qMatrix); // LD
const Picture quantise_transform_np(const Picture &transform, const Array2D &qIndices, const Array1D &qMatrix);
const Picture inverse_quantise_transform_np(const Picture &qCoeffs, const Array2D &qIndices, const Array1D &qMatrix);
const Picture inverse_quantise_transform(const Picture &qCoeffs, const Array2D &qIndices, const Array1D &qMatrix);
#endif
