#include "Quantisation.h"

#include "Hip.h"

namespace {
typedef int (*PlaneOp)(vc2hip_ctx *, const int32_t *, int, int, int, const int32_t *, int, int, const int32_t *, int32_t *);
Array2D planeOp(PlaneOp op, const Array2D &in, const Array2D &qIndices, const Array1D &qMatrix) {
  Array2D out(in.shape());
  vc2hip_ctx *c = hipContext();
  hipCheck(c, op(c, in.data(), (int)in.shape()[0], (int)in.shape()[1], (int)(qMatrix.size() - 1) / 3, qIndices.data(),
                 (int)qIndices.shape()[0], (int)qIndices.shape()[1], qMatrix.data(), out.data()));
  return out;
}
Picture pictureOp(PlaneOp op, const Picture &in, const Array2D &q, const Array1D &m) {
  Picture out(in.format());
  out.y(planeOp(op, in.y(), q, m));
  out.c1(planeOp(op, in.c1(), q, m));
  out.c2(planeOp(op, in.c2(), q, m));
  return out;
}
} // namespace

const Array2D quantise_transform_np(const Array2D &c, const Array2D &q, const Array1D &m) { return planeOp(vc2hip_quantise_np, c, q, m); }
const Array2D inverse_quantise_transform_np(const Array2D &c, const Array2D &q, const Array1D &m) { return planeOp(vc2hip_dequantise_np, c, q, m); }
const Array2D inverse_quantise_transform(const Array2D &c, const Array2D &q, const Array1D &m) { return planeOp(vc2hip_dequantise_ld, c, q, m); }
const Array2D quantise_transform(const Array2D &c, const Array2D &q, const Array1D &m) { return planeOp(vc2hip_quantise_ld, c, q, m); }
const Picture quantise_transform(const Picture &t, const Array2D &q, const Array1D &m) { return pictureOp(vc2hip_quantise_ld, t, q, m); }
const Picture quantise_transform_np(const Picture &t, const Array2D &q, const Array1D &m) { return pictureOp(vc2hip_quantise_np, t, q, m); }
const Picture inverse_quantise_transform_np(const Picture &t, const Array2D &q, const Array1D &m) { return pictureOp(vc2hip_dequantise_np, t, q, m); }
const Picture inverse_quantise_transform(const Picture &t, const Array2D &q, const Array1D &m) { return pictureOp(vc2hip_dequantise_ld, t, q, m); }
