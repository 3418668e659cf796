#include "Slices.h"

#include "Hip.h"

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
