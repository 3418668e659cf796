// Integer lifting DWT / IDWT for the seven VC-2 wavelet kernels, one fused 2-D level per launch.
//
// Replaces waveletTransform / inverseWaveletTransform and their per-kernel level functions
// (/root/reference/src/Library/src/WaveletTransform.cpp:262-342, :478-1265), with waveletPad
// (:79-94), the sample read/convert of Arrays.cpp:333-379, dequantisation (Quantisation.cpp:86-95)
// and clip + sample write (Picture.cpp:284-292, Arrays.cpp:381-426) fused into the first / last
// level.
//
// One workgroup owns a tile of whole slices.  The tile plus a halo is staged in LDS split into
// four parity planes A[row parity][col parity] (== the LL / HL / LH / HH phases), so every lifting
// step, horizontal or vertical, is a unit-stride LDS access across the wavefront, and every
// subband block of a slice is written to the coefficient store as one contiguous run.
// All arithmetic is int32, bit-exact with the reference; no MFMA (integer lifting is not a
// contraction).
#include "vc2hip_internal.h"
#include "vc2hip_wavelets.h"

__constant__ QuantTables c_q;

void vc2_upload_tables(const QuantTables &t, hipStream_t s) {
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(c_q), &t, sizeof t, 0, hipMemcpyHostToDevice, s);
}

// LDS window: four parity planes of WYP x WXP ints
struct Window {
  int *base;
  int wyp, wxp;
  __device__ __forceinline__ int *plane(int rp, int cp) const { return base + (rp * 2 + cp) * wyp * wxp; }
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// one horizontal lifting step on window rows [r0, r1) (pair rows, both row parities)
template <int K, int S, bool INVERSE>
__device__ __forceinline__ void h_step(const Window &w, int kx_base, int npx, int jlo, int jhi) {
  constexpr bool odd = step_targets_odd<K, S>();
  const int rows = 2 * w.wyp, cols = jhi - jlo;
  for (int e = threadIdx.x; e < rows * cols; e += blockDim.x) {
    const int r = e / cols, j = jlo + (e - r * cols);
    int *trow = w.base + ((r / w.wyp) * 2 + (odd ? 1 : 0)) * w.wyp * w.wxp + (r % w.wyp) * w.wxp;
    const int *srow = w.base + ((r / w.wyp) * 2 + (odd ? 0 : 1)) * w.wyp * w.wxp + (r % w.wyp) * w.wxp;
    const int kg = kx_base + j;
    if (kg < 0 || kg >= npx) continue;
    auto at = [&](int d) -> int {
      const int jj = clampi(clampi(kg + d, 0, npx - 1) - kx_base, 0, w.wxp - 1);
      return srow[jj];
    };
    const int dlt = lift_delta<K, S>(at);
    if (INVERSE) trow[j] -= dlt; else trow[j] += dlt;
  }
}

// one vertical lifting step on window columns [jlo, jhi) of both column parities
template <int K, int S, bool INVERSE>
__device__ __forceinline__ void v_step(const Window &w, int ky_base, int npy, int jlo, int jhi) {
  constexpr bool odd = step_targets_odd<K, S>();
  const int cols = jhi - jlo;
  const int items = w.wyp * 2 * cols;
  for (int e = threadIdx.x; e < items; e += blockDim.x) {
    const int i = e / (2 * cols), rem = e - i * 2 * cols;
    const int cp = rem / cols, j = jlo + (rem - cp * cols);
    int *t = w.plane(odd ? 1 : 0, cp);
    const int *s = w.plane(odd ? 0 : 1, cp);
    const int kg = ky_base + i;
    if (kg < 0 || kg >= npy) continue;
    auto at = [&](int d) -> int {
      const int ii = clampi(clampi(kg + d, 0, npy - 1) - ky_base, 0, w.wyp - 1);
      return s[ii * w.wxp + j];
    };
    const int dlt = lift_delta<K, S>(at);
    if (INVERSE) t[i * w.wxp + j] -= dlt; else t[i * w.wxp + j] += dlt;
  }
}

template <int K, int S, bool INVERSE, bool HORIZ>
__device__ __forceinline__ void run_step(const Window &w, int k_base, int np, int jlo, int jhi) {
  if constexpr (HORIZ) h_step<K, S, INVERSE>(w, k_base, np, jlo, jhi);
  else v_step<K, S, INVERSE>(w, k_base, np, jlo, jhi);
  __syncthreads();
}

template <int K, bool INVERSE, bool HORIZ>
__device__ __forceinline__ void run_steps(const Window &w, int k_base, int np, int jlo, int jhi) {
  constexpr int N = WT<K>::nsteps;
  if constexpr (!INVERSE) {
    run_step<K, 0, false, HORIZ>(w, k_base, np, jlo, jhi);
    run_step<K, 1, false, HORIZ>(w, k_base, np, jlo, jhi);
    if constexpr (N == 4) {
      run_step<K, 2, false, HORIZ>(w, k_base, np, jlo, jhi);
      run_step<K, 3, false, HORIZ>(w, k_base, np, jlo, jhi);
    }
  } else {
    if constexpr (N == 4) {
      run_step<K, 3, true, HORIZ>(w, k_base, np, jlo, jhi);
      run_step<K, 2, true, HORIZ>(w, k_base, np, jlo, jhi);
    }
    run_step<K, 1, true, HORIZ>(w, k_base, np, jlo, jhi);
    run_step<K, 0, true, HORIZ>(w, k_base, np, jlo, jhi);
  }
}

// Quantisation.cpp:86-95 with the tables of :40-66 / :78-83
__device__ __forceinline__ int dequant(int v, int aq) {
  if (v == 0) return 0;
  const int qf = c_q.qf[aq], off = c_q.off[aq];
  const unsigned mag = v < 0 ? 0u - (unsigned)v : (unsigned)v;
  int a = (int)(mag * (unsigned)qf);
  if (a > 0) a = (int)((unsigned)a + (unsigned)off);
  a = (int)((unsigned)a + 2u);
  a /= 4;
  return v < 0 ? (int)(0u - (unsigned)a) : a;
}

// ------------------------------------------------------------------------------------------
// forward level
// ------------------------------------------------------------------------------------------
template <int K, bool FIRST>
__global__ __launch_bounds__(256) void k_fwd_level(const LevelParams p) {
  extern __shared__ int lds[];
  const int comp = blockIdx.z % 3, pic = blockIdx.z / 3;
  if ((int)blockIdx.x >= p.tiles_x[comp] || (int)blockIdx.y >= p.tiles_y[comp]) return;
  constexpr int HY = halo_y<K>(), HX = halo_x<K>(), ACC = WT<K>::accuracy;
  const int fh = p.fh[comp], fw = p.fw[comp];
  const int TY = p.tsy[comp] * fh, TX = p.tsx[comp] * fw;
  const int y0 = blockIdx.y * TY, x0 = blockIdx.x * TX;
  const int WY = TY + 2 * HY, WX = TX + 2 * HX;
  Window w{lds, WY / 2, WX / 2};
  const int in_h = p.in_h[comp], in_w = p.in_w[comp];

  // ---- stage tile + halo (fused: sample unpack, offset, edge-replicate padding, accuracy shift)
  for (int e = threadIdx.x; e < WY * WX; e += blockDim.x) {
    const int r = e / WX, c = e - r * WX;
    const int gy = y0 - HY + r, gx = x0 - HX + c;
    int v = 0;
    if (gy >= 0 && gy < in_h && gx >= 0 && gx < in_w) {
      if constexpr (FIRST) {
        const int sy = min(gy, p.pic_h[comp] - 1), sx = min(gx, p.pic_w[comp] - 1);
        const uint8_t *src = (const uint8_t *)p.plane[comp] + (size_t)pic * p.plane_stride[comp] +
                             ((size_t)sy * p.pic_w[comp] + sx) * p.word_bytes;
        unsigned u;
        if (p.word_bytes == 2) {
          const unsigned short h = *(const unsigned short *)src;
          u = ((h & 0xFF) << 8) | (h >> 8);
        } else {
          u = 0;
          for (int b = 0; b < p.word_bytes; ++b) u = (u << 8) | src[b];
        }
        v = (int)(u >> (comp ? p.sample_shift_c : p.sample_shift)) - (comp ? p.sample_offset_c : p.sample_offset);
      } else {
        v = ((const int32_t *)p.plane[comp])[(size_t)pic * p.plane_stride[comp] + (size_t)gy * in_w + gx];
      }
      v = (int)((unsigned)v << ACC);
    }
    w.plane(r & 1, c & 1)[(r >> 1) * w.wxp + (c >> 1)] = v;
  }
  __syncthreads();

  // ---- lifting: horizontal on every window row, then vertical on the core columns
  run_steps<K, false, true>(w, (x0 - HX) / 2, in_w / 2, 0, w.wxp);
  run_steps<K, false, false>(w, (y0 - HY) / 2, in_h / 2, HX / 2, HX / 2 + TX / 2);

  // ---- write the four bands of the core
  const int bsh = fh / 2, bsw = fw / 2, blk = bsh * bsw;
  const int tsx = p.tsx[comp], tsy = p.tsy[comp];
  const int items = tsy * tsx * blk;
  const int s_y0 = blockIdx.y * tsy, s_x0 = blockIdx.x * tsx;
  int32_t *store = (int32_t *)p.store + (size_t)pic * p.store_stride;
  for (int band = 0; band < 4; ++band) {
    const int *src = w.plane(band >> 1, band & 1) + (HY / 2) * w.wxp + HX / 2;
    if (band == 0 && !p.ll_to_store) {
      // compact LL plane for the next level
      int32_t *ll = (int32_t *)p.ll[comp] + (size_t)pic * p.ll_stride[comp];
      const int oh = in_h / 2, ow = in_w / 2, th = TY / 2, tw = TX / 2;
      for (int e = threadIdx.x; e < th * tw; e += blockDim.x) {
        const int i = e / tw, j = e - i * tw;
        const int oy = y0 / 2 + i, ox = x0 / 2 + j;
        if (oy < oh && ox < ow) ll[(size_t)oy * ow + ox] = src[i * w.wxp + j];
      }
      continue;
    }
    // band 0 -> store LL (offset 0); bands 1..3 -> HL, LH, HH of this level.
    // plane(rp,cp): HL = even row / odd col = plane(0,1) = band 1; LH = plane(1,0) = band 2.
    const int off = p.coef_off[comp] + (band == 0 ? 0 : p.band_off[comp] * band);
    for (int e = threadIdx.x; e < items; e += blockDim.x) {
      const int s = e / blk, rem = e - s * blk;
      const int sj = s % tsx, si = s / tsx;
      const int r = rem / bsw, c = rem - r * bsw;
      const int sv = s_y0 + si, sh = s_x0 + sj;
      if (sv < p.ys && sh < p.xs)
        store[(size_t)(sv * p.xs + sh) * p.slice_coefs + off + rem] =
            src[(si * bsh + r) * w.wxp + sj * bsw + c];
    }
  }
}

// ------------------------------------------------------------------------------------------
// inverse level
// ------------------------------------------------------------------------------------------
template <int K, bool FINAL>
__global__ __launch_bounds__(256) void k_inv_level(const LevelParams p) {
  extern __shared__ int lds[];
  const int comp = blockIdx.z % 3, pic = blockIdx.z / 3;
  if ((int)blockIdx.x >= p.tiles_x[comp] || (int)blockIdx.y >= p.tiles_y[comp]) return;
  constexpr int HY = halo_y<K>(), HX = halo_x<K>(), ACC = WT<K>::accuracy;
  const int fh = p.fh[comp], fw = p.fw[comp];
  const int TY = p.tsy[comp] * fh, TX = p.tsx[comp] * fw;
  const int y0 = blockIdx.y * TY, x0 = blockIdx.x * TX;
  const int WY = TY + 2 * HY, WX = TX + 2 * HX;
  Window w{lds, WY / 2, WX / 2};
  const int out_h = p.in_h[comp], out_w = p.in_w[comp]; // plane size at this level
  const int npy = out_h / 2, npx = out_w / 2;
  const int ky_base = (y0 - HY) / 2, kx_base = (x0 - HX) / 2;
  const int bsh = fh / 2, bsw = fw / 2;
  const int32_t *store = (int32_t *)p.store + (size_t)pic * p.store_stride;
  const int32_t *qidx = p.qidx ? p.qidx + (size_t)pic * p.ys * p.xs : nullptr;

  // ---- gather the four bands of tile + halo (fused dequantisation)
  const int wn = w.wyp * w.wxp;
  for (int e = threadIdx.x; e < 4 * wn; e += blockDim.x) {
    const int band = e / wn, rem = e - band * wn;
    const int i = rem / w.wxp, j = rem - i * w.wxp;
    const int by = ky_base + i, bx = kx_base + j;
    int v = 0;
    if (by >= 0 && by < npy && bx >= 0 && bx < npx) {
      if (band == 0 && !p.ll_from_store) {
        v = ((const int32_t *)p.ll[comp])[(size_t)pic * p.ll_stride[comp] + (size_t)by * npx + bx];
      } else {
        const int sv = by / bsh, sh = bx / bsw;
        const int r = by - sv * bsh, c = bx - sh * bsw;
        const int off = p.coef_off[comp] + (band == 0 ? 0 : p.band_off[comp] * band);
        v = store[(size_t)(sv * p.xs + sh) * p.slice_coefs + off + r * bsw + c];
        if (p.dequant) {
          const int q = qidx[sv * p.xs + sh];
          const int m = band == 0 ? p.qmatrix[0] : p.qmatrix[p.band + band - 1];
          const int aq = max(q - m, 0);
          if (aq > 119) atomicOr(p.err, VC2_DEVERR_QINDEX);
          v = dequant(v, min(aq, 119));
        }
      }
    }
    w.plane(band >> 1, band & 1)[rem] = v;
  }
  __syncthreads();

  // ---- inverse lifting: vertical on every window column, then horizontal on the core rows
  run_steps<K, true, false>(w, ky_base, npy, 0, w.wxp);
  // the horizontal pass lifts along x, so it must cover the whole window width (the second step
  // reads first-step results in the halo columns); only the core rows are consumed afterwards
  run_steps<K, true, true>(w, kx_base, npx, 0, w.wxp);

  // ---- interleave, round, and write (FINAL: clip + offset + justify + big-endian words)
  const int lim_h = FINAL ? p.pic_h[comp] : out_h, lim_w = FINAL ? p.pic_w[comp] : out_w;
  for (int e = threadIdx.x; e < TY * TX; e += blockDim.x) {
    const int r = e / TX, c = e - r * TX;
    const int gy = y0 + r, gx = x0 + c;
    if (gy >= lim_h || gx >= lim_w) continue;
    int v = w.plane(r & 1, c & 1)[((r + HY) >> 1) * w.wxp + ((c + HX) >> 1)];
    if (ACC) v = (v + (1 << (ACC - 1))) >> ACC;
    if constexpr (FINAL) {
      v = min(max(v, p.clip_lo), p.clip_hi);
      const unsigned u = (unsigned)(v + p.sample_offset) << p.sample_shift;
      uint8_t *dst = (uint8_t *)p.plane[comp] + (size_t)pic * p.plane_stride[comp] +
                     ((size_t)gy * lim_w + gx) * p.word_bytes;
      if (p.word_bytes == 2) {
        *(unsigned short *)dst = (unsigned short)(((u & 0xFF) << 8) | ((u >> 8) & 0xFF));
      } else {
        for (int b = 0; b < p.word_bytes; ++b) dst[b] = (uint8_t)(u >> (8 * (p.word_bytes - 1 - b)));
      }
    } else {
      ((int32_t *)p.plane[comp])[(size_t)pic * p.plane_stride[comp] + (size_t)gy * out_w + gx] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
template <int K> static size_t lds_bytes_k(const LevelParams &p) {
  size_t m = 0;
  for (int c = 0; c < 3; ++c) {
    const size_t wy = (size_t)p.tsy[c] * p.fh[c] + 2 * halo_y<K>();
    const size_t wx = (size_t)p.tsx[c] * p.fw[c] + 2 * halo_x<K>();
    m = wy * wx * 4 > m ? wy * wx * 4 : m;
  }
  return m;
}

size_t vc2_level_lds_bytes(int kernel, const LevelParams &p) {
  switch (kernel) {
    case VC2HIP_DD97: return lds_bytes_k<VC2HIP_DD97>(p);
    case VC2HIP_LEGALL: return lds_bytes_k<VC2HIP_LEGALL>(p);
    case VC2HIP_DD137: return lds_bytes_k<VC2HIP_DD137>(p);
    case VC2HIP_HAAR0: return lds_bytes_k<VC2HIP_HAAR0>(p);
    case VC2HIP_HAAR1: return lds_bytes_k<VC2HIP_HAAR1>(p);
    case VC2HIP_FIDELITY: return lds_bytes_k<VC2HIP_FIDELITY>(p);
    case VC2HIP_DAUB97: return lds_bytes_k<VC2HIP_DAUB97>(p);
  }
  return 0;
}

int vc2_halo_x(int kernel) {
  switch (kernel) {
    case VC2HIP_DD97: return halo_x<VC2HIP_DD97>();
    case VC2HIP_LEGALL: return halo_x<VC2HIP_LEGALL>();
    case VC2HIP_DD137: return halo_x<VC2HIP_DD137>();
    case VC2HIP_FIDELITY: return halo_x<VC2HIP_FIDELITY>();
    case VC2HIP_DAUB97: return halo_x<VC2HIP_DAUB97>();
    default: return 0;
  }
}
int vc2_halo_y(int kernel) {
  switch (kernel) {
    case VC2HIP_DD97: return halo_y<VC2HIP_DD97>();
    case VC2HIP_LEGALL: return halo_y<VC2HIP_LEGALL>();
    case VC2HIP_DD137: return halo_y<VC2HIP_DD137>();
    case VC2HIP_FIDELITY: return halo_y<VC2HIP_FIDELITY>();
    case VC2HIP_DAUB97: return halo_y<VC2HIP_DAUB97>();
    default: return 0;
  }
}

void vc2_prof_begin(Launcher &L, const char *name, hipStream_t s);
void vc2_prof_end(Launcher &L, hipStream_t s);

template <int K, bool EDGE, bool INV>
static void launch_level(Launcher &L, const LevelParams &p, int n_pictures, hipStream_t s) {
  int gx = 0, gy = 0;
  for (int c = 0; c < 3; ++c) {
    gx = p.tiles_x[c] > gx ? p.tiles_x[c] : gx;
    gy = p.tiles_y[c] > gy ? p.tiles_y[c] : gy;
  }
  const size_t lds = lds_bytes_k<K>(p);
  dim3 grid(gx, gy, 3 * n_pictures), block(256);
  if constexpr (INV) {
    vc2_allow_lds((const void *)k_inv_level<K, EDGE>, 160 * 1024);
    vc2_prof_begin(L, EDGE ? "idwt_level_final" : "idwt_level", s);
    VC2_LAUNCH(L, (k_inv_level<K, EDGE>), grid, block, lds, s, p);
  } else {
    vc2_allow_lds((const void *)k_fwd_level<K, EDGE>, 160 * 1024);
    vc2_prof_begin(L, EDGE ? "dwt_level_first" : "dwt_level", s);
    VC2_LAUNCH(L, (k_fwd_level<K, EDGE>), grid, block, lds, s, p);
  }
  vc2_prof_end(L, s);
}

template <bool INV>
static int dispatch_level(Launcher &L, int kernel, bool edge, const LevelParams &p, int n, hipStream_t s) {
#define VC2_CASE(KK)                                                      \
  case KK:                                                                \
    if (edge) launch_level<KK, true, INV>(L, p, n, s);                    \
    else launch_level<KK, false, INV>(L, p, n, s);                        \
    return 0;
  switch (kernel) {
    VC2_CASE(VC2HIP_DD97)
    VC2_CASE(VC2HIP_LEGALL)
    VC2_CASE(VC2HIP_DD137)
    VC2_CASE(VC2HIP_HAAR0)
    VC2_CASE(VC2HIP_HAAR1)
    VC2_CASE(VC2HIP_FIDELITY)
    VC2_CASE(VC2HIP_DAUB97)
  }
#undef VC2_CASE
  return VC2HIP_EINVAL;
}

int vc2_launch_forward_level(Launcher &L, int kernel, bool first, const LevelParams &p, int n, hipStream_t s) {
  return dispatch_level<false>(L, kernel, first, p, n, s);
}
int vc2_launch_inverse_level(Launcher &L, int kernel, bool final_level, const LevelParams &p, int n, hipStream_t s) {
  return dispatch_level<true>(L, kernel, final_level, p, n, s);
}
