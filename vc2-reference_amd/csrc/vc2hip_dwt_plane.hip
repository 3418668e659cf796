// General-geometry wavelet transform: the fallback for pictures whose slices do not fit the LDS tile of the level
// kernels (the reference admits a slice as large as the whole picture, WaveletTransform.cpp:116-136).
//
// It works the way the reference does (WaveletTransform.cpp:224-342): on the in-place interleaved int32 plane, level by
// level over a view of stride 2^level, every lifting step of every direction as its own pass -- here one kernel launch
// per pass, straight on HBM (a lifting step writes one parity and reads the other, so it is safe in place).  The plane
// then goes to / comes from the coefficient store through the layout conversion kernels of the fine-grained API.
// Slow (a dozen launches per level) and only taken when nothing faster applies; same results.
#include "vc2hip_internal.h"
#include "vc2hip_wavelets.h"

void vc2_prof_begin(Launcher &L, const char *name, hipStream_t s);
void vc2_prof_end(Launcher &L, hipStream_t s);

namespace {

// raw planar words -> padded int32 plane (Arrays.cpp:333-379 + waveletPad, WaveletTransform.cpp:79-94)
__global__ void k_plane_ingest(const uint8_t *raw, long long raw_stride, int pic_h, int pic_w, int word_bytes, int shift, int offset,
                               int32_t *plane, long long plane_stride, int ph, int pw) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, pic = blockIdx.z;
  if (x >= pw) return;
  const uint8_t *q = raw + (size_t)pic * raw_stride + ((size_t)min(y, pic_h - 1) * pic_w + min(x, pic_w - 1)) * word_bytes;
  unsigned u = 0;
  for (int b = 0; b < word_bytes; ++b) u = (u << 8) | q[b];
  plane[(size_t)pic * plane_stride + (size_t)y * pw + x] = (int)(u >> shift) - offset;
}
// int32 plane -> raw planar words, clipped (Picture.cpp:284-292, Arrays.cpp:381-426); the padding is cropped
__global__ void k_plane_emit(const int32_t *plane, long long plane_stride, int pw, uint8_t *raw, long long raw_stride, int pic_h,
                             int pic_w, int word_bytes, int shift, int offset, int lo, int hi) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, pic = blockIdx.z;
  if (x >= pic_w) return;
  const int v = min(max(plane[(size_t)pic * plane_stride + (size_t)y * pw + x], lo), hi);
  const unsigned u = (unsigned)(v + offset) << shift;
  uint8_t *q = raw + (size_t)pic * raw_stride + ((size_t)y * pic_w + x) * word_bytes;
  for (int b = 0; b < word_bytes; ++b) q[b] = (uint8_t)(u >> (8 * (word_bytes - 1 - b)));
}
// the accuracy shift of a level over its view: forward << acc before the level, inverse (x + 2^(acc-1)) >> acc after it
template <bool INV> __global__ void k_plane_shift(int32_t *plane, long long plane_stride, int pw, int level, int vh, int vw, int acc) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, pic = blockIdx.z;
  if (x >= vw || y >= vh) return;
  int32_t *e = plane + (size_t)pic * plane_stride + ((size_t)y << level) * pw + ((size_t)x << level);
  *e = INV ? (*e + (1 << (acc - 1))) >> acc : (int)((unsigned)*e << acc);
}
// lifting step S of wavelet K along rows (VERT = false) or columns of the level's view; one thread per target sample
template <int K, int S, bool INV, bool VERT>
__global__ void k_plane_step(int32_t *plane, long long plane_stride, int pw, int level, int vh, int vw) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y, pic = blockIdx.z; // a: pair index, b: the line
  const int np = (VERT ? vh : vw) >> 1, lines = VERT ? vw : vh;
  if (a >= np || b >= lines) return;
  int32_t *base = plane + (size_t)pic * plane_stride;
  constexpr bool odd = step_targets_odd<K, S>();
  auto elem = [&](int pair, bool o) -> int32_t * { // sample 2 * pair + o of line b
    const int i = 2 * pair + (o ? 1 : 0);
    const size_t y = VERT ? (size_t)i : (size_t)b, x = VERT ? (size_t)b : (size_t)i;
    return base + (y << level) * pw + (x << level);
  };
  const int d = lift_delta<K, S>([&](int t) -> int { return *elem(min(max(a + t, 0), np - 1), !odd); });
  int32_t *tgt = elem(a, odd);
  *tgt = INV ? *tgt - d : *tgt + d;
}

template <int K, int S, bool INV, bool VERT>
void launch_step(Launcher &L, int32_t *plane, long long stride, int pw, int level, int vh, int vw, int n, hipStream_t s) {
  const int np = (VERT ? vh : vw) >> 1, lines = VERT ? vw : vh;
  VC2_LAUNCH(L, (k_plane_step<K, S, INV, VERT>), dim3((np + 127) / 128, lines, n), dim3(128), 0, s, plane, stride, pw, level, vh, vw);
}
template <int K, bool INV, bool VERT>
void launch_pass(Launcher &L, int32_t *plane, long long stride, int pw, int level, int vh, int vw, int n, hipStream_t s) {
  constexpr int N = WT<K>::nsteps;
  if constexpr (!INV) {
    launch_step<K, 0, false, VERT>(L, plane, stride, pw, level, vh, vw, n, s);
    launch_step<K, 1, false, VERT>(L, plane, stride, pw, level, vh, vw, n, s);
    if constexpr (N == 4) { launch_step<K, 2, false, VERT>(L, plane, stride, pw, level, vh, vw, n, s); launch_step<K, 3, false, VERT>(L, plane, stride, pw, level, vh, vw, n, s); }
  } else {
    if constexpr (N == 4) { launch_step<K, 3, true, VERT>(L, plane, stride, pw, level, vh, vw, n, s); launch_step<K, 2, true, VERT>(L, plane, stride, pw, level, vh, vw, n, s); }
    launch_step<K, 1, true, VERT>(L, plane, stride, pw, level, vh, vw, n, s);
    launch_step<K, 0, true, VERT>(L, plane, stride, pw, level, vh, vw, n, s);
  }
}
template <int K> void transform_k(Launcher &L, int32_t *plane, long long stride, int ph, int pw, int depth, bool inverse, int n, hipStream_t s) {
  constexpr int ACC = WT<K>::accuracy;
  if (!inverse) {
    for (int level = 0; level < depth; ++level) { // WaveletTransform.cpp:262-281: shift, horizontal steps, vertical steps
      const int vh = ph >> level, vw = pw >> level;
      if (ACC) VC2_LAUNCH(L, (k_plane_shift<false>), dim3((vw + 127) / 128, vh, n), dim3(128), 0, s, plane, stride, pw, level, vh, vw, ACC);
      launch_pass<K, false, false>(L, plane, stride, pw, level, vh, vw, n, s);
      launch_pass<K, false, true>(L, plane, stride, pw, level, vh, vw, n, s);
    }
  } else {
    for (int level = depth - 1; level >= 0; --level) { // :321-342: vertical steps, horizontal steps, rounding shift
      const int vh = ph >> level, vw = pw >> level;
      launch_pass<K, true, true>(L, plane, stride, pw, level, vh, vw, n, s);
      launch_pass<K, true, false>(L, plane, stride, pw, level, vh, vw, n, s);
      if (ACC) VC2_LAUNCH(L, (k_plane_shift<true>), dim3((vw + 127) / 128, vh, n), dim3(128), 0, s, plane, stride, pw, level, vh, vw, ACC);
    }
  }
}

} // namespace

// n planes of ph x pw (multiples of 2^depth), in place
int vc2_launch_plane_transform(Launcher &L, int kernel, int32_t *plane, long long plane_stride, int ph, int pw, int depth, bool inverse,
                               int n, hipStream_t s) {
  vc2_prof_begin(L, inverse ? "idwt_plane_general" : "dwt_plane_general", s);
  int rc = 0;
  switch (kernel) {
    case VC2HIP_DD97: transform_k<VC2HIP_DD97>(L, plane, plane_stride, ph, pw, depth, inverse, n, s); break;
    case VC2HIP_LEGALL: transform_k<VC2HIP_LEGALL>(L, plane, plane_stride, ph, pw, depth, inverse, n, s); break;
    case VC2HIP_DD137: transform_k<VC2HIP_DD137>(L, plane, plane_stride, ph, pw, depth, inverse, n, s); break;
    case VC2HIP_HAAR0: transform_k<VC2HIP_HAAR0>(L, plane, plane_stride, ph, pw, depth, inverse, n, s); break;
    case VC2HIP_HAAR1: transform_k<VC2HIP_HAAR1>(L, plane, plane_stride, ph, pw, depth, inverse, n, s); break;
    case VC2HIP_FIDELITY: transform_k<VC2HIP_FIDELITY>(L, plane, plane_stride, ph, pw, depth, inverse, n, s); break;
    case VC2HIP_DAUB97: transform_k<VC2HIP_DAUB97>(L, plane, plane_stride, ph, pw, depth, inverse, n, s); break;
    default: rc = VC2HIP_EINVAL;
  }
  vc2_prof_end(L, s);
  return rc;
}
// LD pictures on the whole-plane path: the LL band of the interleaved plane is not the dequantised residuals that
// store_to_plane put there but the DC-predicted reconstruction (inverse_quantise_LLSubband, Quantisation.cpp:287-306),
// which the LL kernels of vc2hip_slices.hip left in the compact LL plane
__global__ void k_ll_into_plane(const int32_t *ll, long long ll_stride, int llh, int llw, int32_t *plane, long long plane_stride, int pw, int depth) {
  const int x = blockIdx.x * 128 + threadIdx.x, y = blockIdx.y, pic = blockIdx.z;
  if (x >= llw || y >= llh) return;
  plane[(size_t)pic * plane_stride + ((size_t)y << depth) * pw + ((size_t)x << depth)] = ll[(size_t)pic * ll_stride + (size_t)y * llw + x];
}
void vc2_launch_ll_into_plane(Launcher &L, const int32_t *ll, long long ll_stride, int llh, int llw, int32_t *plane, long long plane_stride,
                              int pw, int depth, int n, hipStream_t s) {
  vc2_prof_begin(L, "ld_ll_into_plane", s);
  VC2_LAUNCH(L, k_ll_into_plane, dim3((llw + 127) / 128, llh, n), dim3(128), 0, s, ll, ll_stride, llh, llw, plane, plane_stride, pw, depth);
  vc2_prof_end(L, s);
}
void vc2_launch_plane_ingest(Launcher &L, const void *raw, long long raw_stride, int pic_h, int pic_w, int word_bytes, int bit_depth,
                             int32_t *plane, long long plane_stride, int ph, int pw, int n, hipStream_t s) {
  vc2_prof_begin(L, "plane_ingest", s);
  VC2_LAUNCH(L, k_plane_ingest, dim3((pw + 127) / 128, ph, n), dim3(128), 0, s, (const uint8_t *)raw, raw_stride, pic_h, pic_w, word_bytes,
             8 * word_bytes - bit_depth, 1 << (bit_depth - 1), plane, plane_stride, ph, pw);
  vc2_prof_end(L, s);
}
void vc2_launch_plane_emit(Launcher &L, const int32_t *plane, long long plane_stride, int pw, void *raw, long long raw_stride, int pic_h,
                           int pic_w, int word_bytes, int bit_depth, int n, hipStream_t s) {
  vc2_prof_begin(L, "plane_emit", s);
  VC2_LAUNCH(L, k_plane_emit, dim3((pic_w + 127) / 128, pic_h, n), dim3(128), 0, s, plane, plane_stride, pw, (uint8_t *)raw, raw_stride,
             pic_h, pic_w, word_bytes, 8 * word_bytes - bit_depth, 1 << (bit_depth - 1), -(1 << (bit_depth - 1)), (1 << (bit_depth - 1)) - 1);
  vc2_prof_end(L, s);
}
