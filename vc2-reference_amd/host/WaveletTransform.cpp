#include "WaveletTransform.h"

#include <ostream>
#include <stdexcept>

#include "Hip.h"

std::ostream &operator<<(std::ostream &os, WaveletKernel kernel) {
  switch (kernel) {
    case DD97: return os << "Deslauriers-Dubuc (9,7) (\"DD97\")";
    case LeGall: return os << "LeGall (5,3) (\"LeGall\")";
    case DD137: return os << "Deslauriers-Dubuc (13,7) (\"DD137\")";
    case Haar0: return os << "Haar with no shift (\"Haar0\")";
    case Haar1: return os << "Haar with single shift per level (\"Haar1\")";
    case Fidelity: return os << "Fidelity filter (\"Fidelity\")";
    case Daub97: return os << "Daubechies (9,7) integer approximation (\"Daub97\")";
    case NullKernel: return os << "Null kernel (\"NullKernel\")";
  }
  return os << "Unknown wavelet kernel!";
}
WaveletKernel parseWaveletKernel(const std::string &t) {
  if (t == "DD97") return DD97;
  if (t == "LeGall") return LeGall;
  if (t == "DD137") return DD137;
  if (t == "Haar0") return Haar0;
  if (t == "Haar1") return Haar1;
  if (t == "Fidelity") return Fidelity;
  if (t == "Daub97") return Daub97;
  if (t == "NullKernel") return NullKernel;
  throw std::invalid_argument("invalid wavelet kernel");
}

int paddedSize(int size, int depth) { return vc2hip_padded_size(size, depth); }
int sliceSizeIsValid(int d, int l, int c, int n) { return vc2hip_slice_size_is_valid(d, l, c, n); }

const Array1D quantMatrix(WaveletKernel kernel, int depth) {
  if (depth < 0) throw std::domain_error("wavelet depth may not be < 0");
  Array1D q((std::size_t)(3 * depth + 1));
  if (kernel == NullKernel) return q;
  if (vc2hip_quant_matrix((int)kernel, depth, q.data()) != VC2HIP_OK) throw std::invalid_argument("invalid wavelet kernel");
  return q;
}

const Array2D waveletTransform(const Array2D &picture, WaveletKernel kernel, int depth) {
  if (kernel > Daub97) throw std::invalid_argument("invalid wavelet kernel");
  const int h = (int)picture.shape()[0], w = (int)picture.shape()[1];
  Array2D out(paddedSize(h, depth), paddedSize(w, depth));
  vc2hip_ctx *c = hipContext();
  hipCheck(c, vc2hip_dwt_forward(c, picture.data(), h, w, (int)kernel, depth, out.data()));
  return out;
}
const Array2D inverseWaveletTransform(const Array2D &t, WaveletKernel kernel, int depth, Shape2D shape) {
  if (kernel > Daub97) throw std::invalid_argument("invalid wavelet kernel");
  Array2D out(shape);
  vc2hip_ctx *c = hipContext();
  hipCheck(c, vc2hip_dwt_inverse(c, t.data(), (int)t.shape()[0], (int)t.shape()[1], (int)kernel, depth, out.data(),
                                 (int)shape[0], (int)shape[1]));
  return out;
}
const Picture waveletTransform(const Picture &in, WaveletKernel kernel, int depth) {
  const PictureFormat &f = in.format();
  Picture t(PictureFormat(paddedSize(f.lumaHeight(), depth), paddedSize(f.lumaWidth(), depth),
                          paddedSize(f.chromaHeight(), depth), paddedSize(f.chromaWidth(), depth), f.chromaFormat()));
  t.y(waveletTransform(in.y(), kernel, depth));
  t.c1(waveletTransform(in.c1(), kernel, depth));
  t.c2(waveletTransform(in.c2(), kernel, depth));
  return t;
}
const Picture inverseWaveletTransform(const Picture &t, WaveletKernel kernel, int depth, PictureFormat format) {
  Picture p(format);
  p.y(inverseWaveletTransform(t.y(), kernel, depth, format.lumaShape()));
  p.c1(inverseWaveletTransform(t.c1(), kernel, depth, format.chromaShape()));
  p.c2(inverseWaveletTransform(t.c2(), kernel, depth, format.chromaShape()));
  return p;
}
