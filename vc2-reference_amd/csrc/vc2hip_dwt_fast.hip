// Register-blocked DWT / IDWT level kernels (the hot configuration of the generic kernels in
// vc2hip_dwt.hip; same reference semantics, same LevelParams).
//
// Tile = 32 x 128 samples (TY x TX below) of whole slices, anchored inside the plane (the last tile of a row /
// column is shifted back so its core never crosses the plane edge; overlapping cores write
// identical values).  Differences from the generic kernel:
//   * all tile geometry is compile-time (no integer divisions in the inner loops)
//   * 16-byte global loads / stores; 16-byte LDS accesses (ds_read_b128 / ds_write_b128)
//   * each lifting PASS (all steps of one direction) runs in registers: a thread loads a run of
//     4 coefficient pairs plus its halo from the parity planes, applies every step, and writes the
//     4 results after one barrier -- 2 LDS round trips per pass instead of one per tap
//   * plane-edge tap clamping = replication of the edge pair inside the register window
#include <algorithm>

#include "vc2hip_internal.h"
#include "vc2hip_wavelets.h"
#include "vc2hip_store.h"

void vc2_prof_begin(Launcher &L, const char *name, hipStream_t s);
void vc2_prof_end(Launcher &L, hipStream_t s);

__constant__ QuantTables c_qd;
void vc2_upload_tables_fast(const QuantTables &t, hipStream_t s) {
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(c_qd), &t, sizeof t, 0, hipMemcpyHostToDevice, s);
}

#ifndef VC2_FID_NT
#define VC2_FID_NT 512
#endif
#ifdef VC2HIP_STAMPS // diagnostic build only: per-phase times of a tile workgroup
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__device__ unsigned long long *g_tile_stamps;
#define TILE_STAMP(k) do { if (threadIdx.x == 0 && g_tile_stamps) g_tile_stamps[16 * (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) + (k)] = wall_clock64(); } while (0)
#else
#define TILE_STAMP(k)
#endif
namespace {

constexpr int TY = 32, TX = 128;
// threads per workgroup (one tile): the long Fidelity filter works step by step through LDS with a barrier per step and
// is latency-bound -- eight wavefronts per tile halve the work between barriers and double the wavefronts per CU
template <int K> constexpr int NTK = K == VC2HIP_FIDELITY ? VC2_FID_NT : 256;
constexpr int TXQ = TX / 8; // 16-byte chunks (8 samples, 4 pairs) across a tile row

struct I4 {
  int x, y, z, w;
};
__device__ __forceinline__ I4 operator+(I4 a, I4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ I4 operator-(I4 a, I4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
__device__ __forceinline__ I4 operator-(I4 a) { return {-a.x, -a.y, -a.z, -a.w}; }
__device__ __forceinline__ I4 operator+(I4 a, int b) { return {a.x + b, a.y + b, a.z + b, a.w + b}; }
__device__ __forceinline__ I4 operator*(int s, I4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
__device__ __forceinline__ I4 operator>>(I4 a, int s) { return {a.x >> s, a.y >> s, a.z >> s, a.w >> s}; }
__device__ __forceinline__ I4 &operator+=(I4 &a, I4 b) { a = a + b; return a; }
__device__ __forceinline__ I4 &operator-=(I4 &a, I4 b) { a = a - b; return a; }

__device__ __forceinline__ I4 lds_ld4(const int *p) { const int4 v = *(const int4 *)p; return {v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void lds_st4(int *p, I4 v) { *(int4 *)p = make_int4(v.x, v.y, v.z, v.w); }

template <int K> struct Cfg {
  static constexpr int HY = halo_y<K>(), HX = halo_x<K>();
  static constexpr int WY = TY + 2 * HY, WX = TX + 2 * HX, WYP = WY / 2, WXP = WX / 2;
  static constexpr int PADQ = HX / 8;        // halo quads (of pairs) per side, horizontal runs
  static constexpr int NWH = 4 + 8 * PADQ;   // register window of a horizontal run (pairs)
  static constexpr int PADV = HY / 2;        // halo pairs per side, vertical runs
  static constexpr int NWV = 4 + 2 * PADV;
  static constexpr int PLANE = WYP * WXP;    // ints per parity plane
  static constexpr size_t LDS = (size_t)4 * PLANE * 4;
  static constexpr size_t LDS_INV = LDS + 3 * 120 * 4; // + the dequantiser table (factor, offset, domain limit per index)
};

// one lifting step on a register window; E / O are the even / odd parity values of NW pairs
template <int K, int S, bool INV, int NW, class T>
__device__ __forceinline__ void reg_step(T (&E)[NW], T (&O)[NW]) {
  constexpr bool odd = step_targets_odd<K, S>();
  constexpr int dmin = step_dmin<K, S>(), dmax = step_dmax<K, S>();
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    if (i + dmin >= 0 && i + dmax < NW) {
      auto at = [&](int d) -> T { return odd ? E[i + d] : O[i + d]; };
      const T dlt = lift_delta<K, S>(at);
      if constexpr (odd) { if (INV) O[i] -= dlt; else O[i] += dlt; }
      else { if (INV) E[i] -= dlt; else E[i] += dlt; }
    }
  }
}
// replicate the plane-edge pair into the out-of-plane part of the window (tap clamping)
template <int NW, class T> __device__ __forceinline__ void replicate(T (&A)[NW], int nl, int nr) {
#pragma unroll
  for (int i = NW - 2; i >= 0; --i) if (i < nl) A[i] = A[i + 1];
#pragma unroll
  for (int i = 1; i < NW; ++i) if (i > nr) A[i] = A[i - 1];
}
template <int K, int S, bool INV, int NW, class T>
__device__ __forceinline__ void reg_step_edge(T (&E)[NW], T (&O)[NW], bool edge, int nl, int nr) {
  reg_step<K, S, INV>(E, O);
  if (edge) { if (step_targets_odd<K, S>()) replicate(O, nl, nr); else replicate(E, nl, nr); }
}
template <int K, bool INV, int NW, class T>
__device__ __forceinline__ void reg_pass(T (&E)[NW], T (&O)[NW], bool edge, int nl, int nr) {
  if (edge) { replicate(E, nl, nr); replicate(O, nl, nr); }
  constexpr int N = WT<K>::nsteps;
  if constexpr (!INV) {
    reg_step_edge<K, 0, false>(E, O, edge, nl, nr);
    reg_step_edge<K, 1, false>(E, O, edge, nl, nr);
    if constexpr (N == 4) { reg_step_edge<K, 2, false>(E, O, edge, nl, nr); reg_step_edge<K, 3, false>(E, O, edge, nl, nr); }
  } else {
    if constexpr (N == 4) { reg_step_edge<K, 3, true>(E, O, edge, nl, nr); reg_step_edge<K, 2, true>(E, O, edge, nl, nr); }
    reg_step_edge<K, 1, true>(E, O, edge, nl, nr);
    reg_step_edge<K, 0, true>(E, O, edge, nl, nr);
  }
}

// horizontal pass over rows [row_lo, row_hi) of both row parities, core columns only.
// NIT = iterations of 256 threads; results are held in registers across the barrier.
template <int K, bool INV, int NIT>
__device__ __forceinline__ void h_pass(int *lds, int row_lo, int n_rows, int kx_base, int npx) {
  constexpr int NT = NTK<K>;
  using C = Cfg<K>;
  constexpr int NW = C::NWH, P = 4 * C::PADQ;
  I4 re[NIT], ro[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = it * NT + threadIdx.x;
    if (id < n_rows * TXQ) {
      const int rr = row_lo + id / TXQ, t = id % TXQ; // rr indexes rows of the (2*WYP)-row stack
      const int j0 = C::HX / 2 + 4 * t;
      const int *erow = lds + (rr / C::WYP * 2 + 0) * C::PLANE + (rr % C::WYP) * C::WXP + j0 - P;
      const int *orow = erow + C::PLANE;
      int E[NW], O[NW];
#pragma unroll
      for (int q = 0; q < NW / 4; ++q) {
        const I4 a = lds_ld4(erow + 4 * q), b = lds_ld4(orow + 4 * q);
        E[4 * q] = a.x; E[4 * q + 1] = a.y; E[4 * q + 2] = a.z; E[4 * q + 3] = a.w;
        O[4 * q] = b.x; O[4 * q + 1] = b.y; O[4 * q + 2] = b.z; O[4 * q + 3] = b.w;
      }
      const int k0 = kx_base + j0 - P; // plane pair index of window entry 0
      const int nl = -k0, nr = npx - 1 - k0;
      reg_pass<K, INV>(E, O, nl > 0 || nr < NW - 1, nl, nr);
      re[it] = {E[P], E[P + 1], E[P + 2], E[P + 3]};
      ro[it] = {O[P], O[P + 1], O[P + 2], O[P + 3]};
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = it * NT + threadIdx.x;
    if (id < n_rows * TXQ) {
      const int rr = row_lo + id / TXQ, t = id % TXQ;
      int *erow = lds + (rr / C::WYP * 2 + 0) * C::PLANE + (rr % C::WYP) * C::WXP + C::HX / 2 + 4 * t;
      lds_st4(erow, re[it]);
      lds_st4(erow + C::PLANE, ro[it]);
    }
  }
  __syncthreads();
}

// vertical pass over the core row pairs, column quads [cq_lo, cq_lo + n_cq) of both column parities
template <int K, bool INV, int NIT>
__device__ __forceinline__ void v_pass(int *lds, int cq_lo, int n_cq, int ky_base, int npy) {
  constexpr int NT = NTK<K>;
  using C = Cfg<K>;
  constexpr int NW = C::NWV, P = C::PADV;
  I4 re[NIT][4], ro[NIT][4];
  const int items = n_cq * 2 * (TY / 8);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = it * NT + threadIdx.x;
    if (id < items) {
      const int cq = cq_lo + id % n_cq, rest = id / n_cq;
      const int cp = rest & 1, rr = rest >> 1;
      const int i0 = C::HY / 2 + 4 * rr;
      const int *ep = lds + (0 * 2 + cp) * C::PLANE + (i0 - P) * C::WXP + 4 * cq;
      const int *op = ep + 2 * C::PLANE;
      I4 E[NW], O[NW];
#pragma unroll
      for (int k = 0; k < NW; ++k) { E[k] = lds_ld4(ep + k * C::WXP); O[k] = lds_ld4(op + k * C::WXP); }
      const int k0 = ky_base + i0 - P;
      const int nl = -k0, nr = npy - 1 - k0;
      reg_pass<K, INV>(E, O, nl > 0 || nr < NW - 1, nl, nr);
#pragma unroll
      for (int k = 0; k < 4; ++k) { re[it][k] = E[P + k]; ro[it][k] = O[P + k]; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = it * NT + threadIdx.x;
    if (id < items) {
      const int cq = cq_lo + id % n_cq, rest = id / n_cq;
      const int cp = rest & 1, rr = rest >> 1;
      const int i0 = C::HY / 2 + 4 * rr;
      int *ep = lds + (0 * 2 + cp) * C::PLANE + i0 * C::WXP + 4 * cq;
      int *op = ep + 2 * C::PLANE;
#pragma unroll
      for (int k = 0; k < 4; ++k) { lds_st4(ep + k * C::WXP, re[it][k]); lds_st4(op + k * C::WXP, ro[it][k]); }
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Step-wise passes for long filters (Fidelity: 8 taps, 14 samples of halo).  The register-window passes above
// would need a window of 20 pairs per parity (>180 VGPRs, 2 wavefronts per SIMD) and compute every lifting step
// on the whole shrinking window; here every lifting step is its own sweep over LDS: a thread updates one quad of
// the target parity in place from the (unchanged) opposite parity, each position exactly once, one barrier
// per step.  Plane-edge tap clamping = clamped read index.
// ------------------------------------------------------------------------------------------
template <int K> constexpr bool stepwise() { return K == VC2HIP_FIDELITY; }
constexpr int floor_div4(int v) { return v >= 0 ? v / 4 : -((-v + 3) / 4); }

// lifting step S along x on rows [row_lo, row_lo + n_rows) of the (2 * WYP)-row stack, pair quads [q_lo, q_hi)
template <int K, int S, bool INV>
__device__ __forceinline__ void step_h(int *lds, int row_lo, int n_rows, int q_lo, int q_hi, int kx_base, int npx) {
  constexpr int NT = NTK<K>;
  using C = Cfg<K>;
  constexpr bool odd = step_targets_odd<K, S>();
  constexpr int dmin = step_dmin<K, S>(), dmax = step_dmax<K, S>();
  constexpr int QL = floor_div4(dmin), QH = floor_div4(3 + dmax), NWIN = (QH - QL + 1) * 4;
  const int nq = q_hi - q_lo, items = n_rows * nq;
  for (int id = threadIdx.x; id < items; id += NT) {
    const int r = id / nq, t = q_lo + (id - r * nq), rr = row_lo + r;
    int *base = lds + (rr / C::WYP * 2) * C::PLANE + (rr % C::WYP) * C::WXP;
    int *own = base + (odd ? C::PLANE : 0) + 4 * t;
    const int *opp = base + (odd ? 0 : C::PLANE);
    int Wn[NWIN];
    const int c0 = 4 * (t + QL);           // window column of Wn[0]
    const int k0 = kx_base + c0;           // its plane pair index
    if (k0 >= 0 && k0 + NWIN - 1 <= npx - 1) {
#pragma unroll
      for (int q = 0; q < NWIN / 4; ++q) {
        const I4 a = lds_ld4(opp + c0 + 4 * q);
        Wn[4 * q] = a.x; Wn[4 * q + 1] = a.y; Wn[4 * q + 2] = a.z; Wn[4 * q + 3] = a.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < NWIN; ++e) Wn[e] = opp[min(max(k0 + e, 0), npx - 1) - kx_base];
    }
    const I4 o = lds_ld4(own);
    int v[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      auto at = [&](int d) -> int { return Wn[i + d - 4 * QL]; };
      const int dlt = lift_delta<K, S>(at);
      if (INV) v[i] -= dlt; else v[i] += dlt;
    }
    lds_st4(own, {v[0], v[1], v[2], v[3]});
  }
  __syncthreads();
}

// lifting step S along y on row pairs [r_lo, r_hi), column quads [cq_lo, cq_lo + n_cq) of both column parities
template <int K, int S, bool INV>
__device__ __forceinline__ void step_v(int *lds, int r_lo, int r_hi, int cq_lo, int n_cq, int ky_base, int npy) {
  constexpr int NT = NTK<K>;
  using C = Cfg<K>;
  constexpr bool odd = step_targets_odd<K, S>();
  constexpr int dmin = step_dmin<K, S>(), dmax = step_dmax<K, S>();
  constexpr int NR = 4 + dmax - dmin;
  const int groups = (r_hi - r_lo + 3) / 4, items = groups * n_cq * 2;
  for (int id = threadIdx.x; id < items; id += NT) {
    const int cq = cq_lo + id % n_cq, rest = id / n_cq;
    const int cp = rest & 1, i0 = r_lo + 4 * (rest >> 1);
    int *own = lds + ((odd ? 1 : 0) * 2 + cp) * C::PLANE + 4 * cq;
    const int *opp = lds + ((odd ? 0 : 1) * 2 + cp) * C::PLANE + 4 * cq;
    I4 Wn[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) {
      const int row = min(max(ky_base + i0 + dmin + e, 0), npy - 1) - ky_base; // clamped tap row (window coordinates)
      Wn[e] = lds_ld4(opp + row * C::WXP);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i0 + i < r_hi) {
        auto at = [&](int d) -> I4 { return Wn[i + d - dmin]; };
        const I4 dlt = lift_delta<K, S>(at);
        I4 o = lds_ld4(own + (i0 + i) * C::WXP);
        if (INV) o -= dlt; else o += dlt;
        lds_st4(own + (i0 + i) * C::WXP, o);
      }
    }
  }
  __syncthreads();
}

// forward: S0 then S1 (the second step's reach decides how far beyond the core the first must be computed);
// inverse: S1 then S0
template <int K, bool INV>
__device__ __forceinline__ void steps_h(int *lds, int row_lo, int n_rows, int kx_base, int npx) {
  using C = Cfg<K>;
  constexpr int NQ = C::WXP / 4, HQ = C::HX / 8; // quads per row, halo quads per side
  static_assert(WT<K>::nsteps == 2 && HQ >= 2, "step-wise passes: two steps, one quad of reach each");
  if constexpr (!INV) {
    step_h<K, 0, false>(lds, row_lo, n_rows, HQ - 1, NQ - HQ + 1, kx_base, npx);
    step_h<K, 1, false>(lds, row_lo, n_rows, HQ, NQ - HQ, kx_base, npx);
  } else {
    step_h<K, 1, true>(lds, row_lo, n_rows, HQ - 1, NQ - HQ + 1, kx_base, npx);
    step_h<K, 0, true>(lds, row_lo, n_rows, HQ, NQ - HQ, kx_base, npx);
  }
}
template <int K, bool INV>
__device__ __forceinline__ void steps_v(int *lds, int cq_lo, int n_cq, int ky_base, int npy) {
  using C = Cfg<K>;
  constexpr int R0 = C::HY / 2, R1 = C::HY / 2 + TY / 2; // core row pairs
  constexpr int FIRST = INV ? 1 : 0, SECOND = INV ? 0 : 1;
  constexpr int lo = R0 + step_dmin<K, SECOND>(), hi = R1 + step_dmax<K, SECOND>();
  static_assert(lo + step_dmin<K, FIRST>() >= 0 && hi - 1 + step_dmax<K, FIRST>() <= C::WYP - 1, "halo too small");
  step_v<K, FIRST, INV>(lds, lo, hi, cq_lo, n_cq, ky_base, npy);
  step_v<K, SECOND, INV>(lds, R0, R1, cq_lo, n_cq, ky_base, npy);
}

__device__ __forceinline__ int dequant_f(int v, int qf, int off) {
  if (v == 0) return 0;
  const unsigned mag = v < 0 ? 0u - (unsigned)v : (unsigned)v;
  int a = (int)(mag * (unsigned)qf);
  if (a > 0) a = (int)((unsigned)a + (unsigned)off);
  a = (int)((unsigned)a + 2u);
  a /= 4;
  return v < 0 ? (int)(0u - (unsigned)a) : a;
}

__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }
constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so
// workgroup `lin` runs on XCD lin % 8.  Re-number the tiles so that every XCD gets one contiguous,
// x-major range of them: horizontally and vertically adjacent tiles, which share halo rows / columns
// and the cache lines at tile edges, then hit the same L2 instead of fetching those lines once per XCD.
// (Placement only changes speed, never results.)
__device__ __forceinline__ bool tile_of_block(int tiles_x, int tiles_y, int &tx, int &ty) {
  const int n = tiles_x * tiles_y;
  const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
  if (lin >= n) return false;
  const int c = lin & 7, k = lin >> 3, chunk = n >> 3, rem = n & 7;
  const int L = c * chunk + min(c, rem) + k;
  ty = L / tiles_x;
  tx = L - ty * tiles_x;
  return true;
}

// ------------------------------------------------------------------------------------------
// forward, deep levels: the slice-by-slice write with the block shape known at compile time (the counterpart of
// gather_slices further down: a thread assembles a slice's [LL |] HL | LH | HH run from whole block rows of the four
// planes and stores it quad by quad; slice row / column of a thread by shifts)
// ------------------------------------------------------------------------------------------
#ifndef VC2_FWD_SHAPED
#define VC2_FWD_SHAPED 1 // 0: the generic loop for every shape (A/B on one box)
#endif
template <class C, int NT, class ST, int LW, int LN, bool LLF>
__device__ __forceinline__ void scatter_slices(const int *core, const LevelParams &p, int comp, ST *store, int32_t *wide,
                                               int s_y0, int s_x0, int chunk0) {
  using S_ = St<ST>;
  constexpr int BSW = 1 << LW, BN = 1 << LN, BSH = BN / BSW;
  constexpr int B0 = LLF ? 0 : 1, NB = 4 - B0, CN = NB * BN, NQ = CN / 4;
  static_assert(CN % 4 == 0 && LN >= LW, "whole quads");
  constexpr int LTSX = ilog2c(TX / 2) - LW, NSL = ((TY / 2) * (TX / 2)) >> LN; // slices across a tile (log2), slices of a tile
  const int rs = p.rec_stride[comp], xs = p.xs;
  for (int sidx = threadIdx.x; sidx < NSL; sidx += NT) {
    const int si = sidx >> LTSX, sj = sidx & ((1 << LTSX) - 1);
    int v[CN];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < BSH; ++r) {
        const int *s = core + (B0 + b) * C::PLANE + (si * BSH + r) * C::WXP + sj * BSW;
        int *d = v + b * BN + r * BSW;
        if constexpr (BSW == 4) { const I4 t = lds_ld4(s); d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
        else if constexpr (BSW == 2) { const int2 t = *(const int2 *)s; d[0] = t.x; d[1] = t.y; }
        else d[0] = s[0];
      }
    const size_t at = (size_t)((s_y0 + si) * xs + s_x0 + sj) * rs + chunk0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) S_::store4(store + at + 4 * k, wide + at + 4 * k, v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
  }
}

// ------------------------------------------------------------------------------------------
// forward level
// ------------------------------------------------------------------------------------------
template <int K, bool FIRST, class ST>
__global__ __launch_bounds__(NTK<K>) void k_fwd_fast(const LevelParams p) {
  constexpr int NT = NTK<K>;
  using S_ = St<ST>;
  using C = Cfg<K>;
  extern __shared__ __attribute__((aligned(16))) int lds[];
  const int comp = blockIdx.z % 3, pic = blockIdx.z / 3;
  int tile_x, tile_y;
  if (!tile_of_block(p.tiles_x[comp], p.tiles_y[comp], tile_x, tile_y)) return;
  TILE_STAMP(0);
  constexpr int HY = C::HY, HX = C::HX, WX = C::WX, WY = C::WY, WXP = C::WXP, ACC = WT<K>::accuracy;
  const int in_h = p.in_h[comp], in_w = p.in_w[comp];
  const int y0 = min(tile_y * TY, in_h - TY), x0 = min(tile_x * TX, in_w - TX);

  // ---- stage tile + halo: 8 samples per item, split into even / odd column planes.
  // All 16-byte loads of a thread are issued first (NLD in flight), then converted: one memory
  // latency per tile instead of one per item.
  {
    const int pic_h = p.pic_h[comp], pic_w = p.pic_w[comp];
    const bool vec_ok = FIRST ? (p.word_bytes == 2 && (pic_w & 7) == 0) : ((in_w & 7) == 0);
    constexpr int NLD = (WY * (WX / 8) + NT - 1) / NT;
    uint4 va[NLD], vb[NLD]; // FIRST: va = 8 samples; else va (, vb) = 8 coefficients of the level plane
    const ST *lvl = FIRST ? nullptr : (const ST *)p.plane[comp] + (size_t)pic * p.plane_stride[comp];
    const int32_t *lvl_w = (FIRST || !S_::narrow) ? nullptr : p.plane_wide[comp] + (size_t)pic * p.plane_stride[comp];
    int kind[NLD];          // 0 skip, 1 vector data loaded, 2 element-wise path
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int id = it * NT + threadIdx.x;
      kind[it] = 0;
      va[it] = make_uint4(0, 0, 0, 0);
      vb[it] = make_uint4(0, 0, 0, 0);
      if (id >= WY * (WX / 8) || VC2_SKIP(p, 1)) continue;
      const int r = id / (WX / 8), ch = id - r * (WX / 8);
      const int gy = y0 - HY + r, gx0 = x0 - HX + 8 * ch;
      if (gy < 0 || gy >= in_h || gx0 + 8 <= 0 || gx0 >= in_w) continue;
      if constexpr (FIRST) {
        if (vec_ok && gx0 >= 0 && gx0 + 8 <= pic_w) {
          const uint8_t *row = (const uint8_t *)p.plane[comp] + (size_t)pic * p.plane_stride[comp] +
                               (size_t)min(gy, pic_h - 1) * pic_w * 2;
          va[it] = *(const uint4 *)(row + (size_t)gx0 * 2);
          kind[it] = 1;
        } else kind[it] = 2;
      } else {
        if (vec_ok && gx0 >= 0 && gx0 + 8 <= in_w) {
          const ST *row = lvl + (size_t)gy * in_w;
          va[it] = *(const uint4 *)(row + gx0);
          if constexpr (!S_::narrow) vb[it] = *(const uint4 *)(row + gx0 + 4);
          kind[it] = 1;
        } else kind[it] = 2;
      }
    }
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      if (kind[it] == 0) continue;
      const int id = it * NT + threadIdx.x;
      const int r = id / (WX / 8), ch = id - r * (WX / 8);
      const int gy = y0 - HY + r, gx0 = x0 - HX + 8 * ch;
      int s[8];
      if constexpr (FIRST) {
        if (kind[it] == 1) {
          const unsigned wv[4] = {va[it].x, va[it].y, va[it].z, va[it].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned b = __builtin_bswap32(wv[k]);
            s[2 * k] = (int)(b >> 16);
            s[2 * k + 1] = (int)(b & 0xFFFFu);
          }
        } else {
          const uint8_t *row = (const uint8_t *)p.plane[comp] + (size_t)pic * p.plane_stride[comp] +
                               (size_t)min(gy, pic_h - 1) * pic_w * p.word_bytes;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int sx = min(max(gx0 + k, 0), pic_w - 1);
            const uint8_t *q = row + (size_t)sx * p.word_bytes;
            unsigned u = 0;
            for (int b = 0; b < p.word_bytes; ++b) u = (u << 8) | q[b];
            s[k] = (int)u;
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          s[k] = (int)((unsigned)((int)((unsigned)s[k] >> (comp ? p.sample_shift_c : p.sample_shift)) - (comp ? p.sample_offset_c : p.sample_offset)) << ACC);
      } else {
        if (kind[it] == 1) {
          if constexpr (S_::narrow) S_::unpack8(va[it], lvl_w + (size_t)gy * in_w + gx0, s);
          else {
            s[0] = (int)va[it].x; s[1] = (int)va[it].y; s[2] = (int)va[it].z; s[3] = (int)va[it].w;
            s[4] = (int)vb[it].x; s[5] = (int)vb[it].y; s[6] = (int)vb[it].z; s[7] = (int)vb[it].w;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const size_t at = (size_t)gy * in_w + min(max(gx0 + k, 0), in_w - 1);
            s[k] = S_::load1(lvl + at, lvl_w + at);
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] = (int)((unsigned)s[k] << ACC);
      }
      int *e = lds + ((r & 1) * 2 + 0) * C::PLANE + (r >> 1) * WXP + 4 * ch;
      lds_st4(e, {s[0], s[2], s[4], s[6]});
      lds_st4(e + C::PLANE, {s[1], s[3], s[5], s[7]});
    }
  }
  __syncthreads();
  TILE_STAMP(1);

  // ---- lifting in registers: horizontal over all in-plane window rows, vertical over the core
  {
    // window rows gy in [0,in_h): stack index rr = rp*WYP + i, row r = 2*i + rp.  Rows outside the
    // plane are skipped (the vertical pass replicates across the plane edge itself).
    constexpr int NITH = (2 * C::WYP * TXQ + NT - 1) / NT;
    if (!VC2_SKIP(p, 2)) {
    if constexpr (stepwise<K>()) {
      steps_h<K, false>(lds, 0, 2 * C::WYP, (x0 - HX) / 2, in_w / 2);
      steps_v<K, false>(lds, HX / 8, TXQ, (y0 - HY) / 2, in_h / 2);
    } else {
      h_pass<K, false, NITH>(lds, 0, 2 * C::WYP, (x0 - HX) / 2, in_w / 2);
      v_pass<K, false, 1>(lds, HX / 8, TXQ, (y0 - HY) / 2, in_h / 2);
    }
    }
  }

  TILE_STAMP(2);
  // ---- write the four bands of the core
  const int fh = p.fh[comp], fw = p.fw[comp];
  const int bsh = fh >> 1, bsw = fw >> 1;
  const int lbsw = ilog2(bsw), lblk = ilog2(bsh) + lbsw;
  const int tsx_l = ilog2(TX / fw);           // slices per tile row (log2)
  const int s_y0 = y0 / fh, s_x0 = x0 / fw;
  ST *store = (ST *)p.store + (size_t)pic * p.store_stride;
  int32_t *wide = S_::narrow ? p.store_wide + (size_t)pic * p.store_stride : nullptr;
  const int *core = lds + (HY / 2) * WXP + HX / 2;
  if (VC2_SKIP(p, 4)) return;
  // Deep levels (band blocks narrower than four coefficients): the level's bands of one slice are one short contiguous
  // run of its record ([LL |] HL | LH | HH), written slice by slice with 16-byte stores assembled from the planes.
  const bool by_slice = lbsw < 2;
  if (by_slice) {
    const int band_n = bsh * bsw, lbn = lblk;
    const int band_first = p.ll_to_store ? 0 : 1;
    const int chunk0 = p.coef_off[comp] + (p.ll_to_store ? 0 : p.band_off[comp]);
    const int chunk_n = (4 - band_first) * band_n, nq = (chunk_n + 3) >> 2;
    const int tsx = TX / fw, nsl = (TY / fh) * tsx;
    const bool al = ((chunk0 | p.rec_stride[comp]) & 3) == 0;
    bool shaped = false;
    if (VC2_FWD_SHAPED && al) {
#define VC2_SHAPE(LW_, LN_, LLF_)                                                                  \
  if (!shaped && lbsw == LW_ && lbn == LN_ && band_first == (LLF_ ? 0 : 1)) {                      \
    scatter_slices<C, NT, ST, LW_, LN_, LLF_>(core, p, comp, store, wide, s_y0, s_x0, chunk0);     \
    shaped = true;                                                                                 \
  }
      VC2_SHAPE(0, 0, true) VC2_SHAPE(0, 1, true) VC2_SHAPE(1, 1, true) VC2_SHAPE(1, 2, true)
      VC2_SHAPE(1, 2, false) VC2_SHAPE(1, 3, false)
#undef VC2_SHAPE
    }
    for (int id = threadIdx.x; id < (shaped ? 0 : nsl * nq); id += NT) {
      const int sidx = id / nq, qd = id - sidx * nq;
      const int si = sidx / tsx, sj = sidx - si * tsx;
      int e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int idx = min(4 * qd + k, chunk_n - 1);
        const int band = band_first + (idx >> lbn), rem = idx & (band_n - 1);
        e[k] = core[band * C::PLANE + ((si << (lblk - lbsw)) + (rem >> lbsw)) * WXP + (sj << lbsw) + (rem & (bsw - 1))];
      }
      const size_t at = (size_t)((s_y0 + si) * p.xs + s_x0 + sj) * p.rec_stride[comp] + chunk0 + 4 * qd;
      if (al && 4 * qd + 4 <= chunk_n) S_::store4(store + at, wide + at, e[0], e[1], e[2], e[3]);
      else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (4 * qd + k < chunk_n) S_::store1(store + at + k, wide + at + k, e[k]);
      }
    }
  }
#pragma unroll 1
  for (int band = 0; band < 4; ++band) {
    const int *src = core + band * C::PLANE;
    if (by_slice && !(band == 0 && !p.ll_to_store)) continue; // written above
    if (band == 0 && !p.ll_to_store) {
      ST *ll = (ST *)p.ll[comp] + (size_t)pic * p.ll_stride[comp];
      int32_t *ll_w = S_::narrow ? p.ll_wide[comp] + (size_t)pic * p.ll_stride[comp] : nullptr;
      const int ow = in_w >> 1;
      const bool v4 = (ow & 3) == 0;
      for (int id = threadIdx.x; id < (TY / 2) * (TX / 8); id += NT) {
        const int i = id / TXQ, jq = id % TXQ;
        const I4 v = lds_ld4(src + i * WXP + 4 * jq);
        const size_t at = (size_t)(y0 / 2 + i) * ow + x0 / 2 + 4 * jq;
        if (v4) S_::store4(ll + at, ll_w + at, v.x, v.y, v.z, v.w);
        else { S_::store1(ll + at, ll_w + at, v.x); S_::store1(ll + at + 1, ll_w + at + 1, v.y);
               S_::store1(ll + at + 2, ll_w + at + 2, v.z); S_::store1(ll + at + 3, ll_w + at + 3, v.w); }
      }
      continue;
    }
    const int off = p.coef_off[comp] + (band == 0 ? 0 : p.band_off[comp] * band);
    if (lbsw >= 2) { // 16-byte stores: a quad never crosses a block row
      for (int id = threadIdx.x; id < (TY / 2) * (TX / 8); id += NT) {
        const int e = id << 2;
        const int s = e >> lblk, rem = e & ((1 << lblk) - 1);
        const int si = s >> tsx_l, sj = s & ((1 << tsx_l) - 1);
        const int r = rem >> lbsw, c = rem & (bsw - 1);
        const I4 v = lds_ld4(src + ((si << (lblk - lbsw)) + r) * WXP + (sj << lbsw) + c);
        const size_t at = (size_t)((s_y0 + si) * p.xs + s_x0 + sj) * p.rec_stride[comp] + off + rem;
        S_::store4(store + at, wide + at, v.x, v.y, v.z, v.w);
      }
    } else {
      for (int e = threadIdx.x; e < (TY / 2) * (TX / 2); e += NT) {
        const int s = e >> lblk, rem = e & ((1 << lblk) - 1);
        const int si = s >> tsx_l, sj = s & ((1 << tsx_l) - 1);
        const int r = rem >> lbsw, c = rem & (bsw - 1);
        const size_t at = (size_t)((s_y0 + si) * p.xs + s_x0 + sj) * p.rec_stride[comp] + off + rem;
        S_::store1(store + at, wide + at, src[((si << (lblk - lbsw)) + r) * WXP + (sj << lbsw) + c]);
      }
    }
  }
  TILE_STAMP(3);
}

#ifndef VC2_INV_SHAPED
#define VC2_INV_SHAPED 1 // 0: the generic loop for every shape (A/B on one box)
#endif
// ------------------------------------------------------------------------------------------
// inverse, deep levels: the slice-by-slice gather with the block shape known at compile time
// ------------------------------------------------------------------------------------------
// A slice's coefficients of the level are CN = (3 or 4) * BN contiguous elements ([LL |] HL | LH | HH, blocks of
// BSH x BSW = 2^(LN-LW) x 2^LW).  One thread takes whole slices: every load of its GB slices (the quads of
// coefficients, the quantiser index) is issued before the first is consumed -- one trip to memory per workgroup for
// every shape of the BASELINE formats -- and band, block row and column of every element are constants: a band's
// quantiser constants are looked up once per slice, a block row goes to LDS as one 4 / 8 / 16-byte write.  (The generic
// loop of k_inv_fast derives all of that per piece of four elements: 145 instructions a piece; its phase was 14 of a
// workgroup's 21 us at the deepest level of UHD, 8 of 15 at the level above.)  Same values as the generic loop.
template <class C, int NT, class ST, int LW, int LN, bool LLF>
__device__ __forceinline__ void gather_slices(int *lds, const int *qtab, const LevelParams &p, int comp, const ST *store,
                                              const int32_t *wide, const int32_t *qidx, int ky_base, int kx_base, int npy,
                                              int npx, int chunk0) {
  using S_ = St<ST>;
  constexpr int BSW = 1 << LW, BN = 1 << LN, BSH = BN / BSW, LH = LN - LW;
  constexpr int B0 = LLF ? 0 : 1, NB = 4 - B0, CN = NB * BN, NQ = CN / 4;
  static_assert(CN % 4 == 0 && LH >= 0, "whole quads");
  constexpr int GB = NQ >= 6 ? 1 : NQ >= 3 ? 2 : NQ == 2 ? 4 : 8;
  constexpr int WYP = C::WYP, WXP = C::WXP;
  const int sr0 = max(ky_base, 0) >> LH, sr1 = min(ky_base + WYP - 1, npy - 1) >> LH;
  const int sc0 = max(kx_base, 0) >> LW, sc1 = min(kx_base + WXP - 1, npx - 1) >> LW;
  const int nsc = sc1 - sc0 + 1, nsl = (sr1 - sr0 + 1) * nsc;
  const unsigned mg_nsc = 0xFFFFFFFFu / (unsigned)nsc + 1u;
  const int rs = p.rec_stride[comp], xs = p.xs;
  int qm[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) qm[b] = (B0 + b) == 0 ? p.qmatrix[0] : p.qmatrix[p.band + B0 + b - 1];
  for (int id0 = threadIdx.x; id0 < nsl; id0 += NT * GB) {
    int4 raw[GB][NQ]; // int16 store: four elements in .x, .y
    int q[GB], sv_[GB], sh_[GB];
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      const int id = id0 + g * NT;
      q[g] = 0; sv_[g] = -1; sh_[g] = 0;
#pragma unroll
      for (int k = 0; k < NQ; ++k) raw[g][k] = make_int4(0, 0, 0, 0);
      if (id >= nsl) continue;
      const int sr = nsc == 1 ? id : (int)__umulhi((unsigned)id, mg_nsc);
      const int sv = sr0 + sr, sh = sc0 + (id - sr * nsc);
      sv_[g] = sv; sh_[g] = sh;
      const ST *src = store + (size_t)(sv * xs + sh) * rs + chunk0;
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        if constexpr (S_::narrow) { const uint2 v = *(const uint2 *)(src + 4 * k); raw[g][k].x = (int)v.x; raw[g][k].y = (int)v.y; }
        else raw[g][k] = *(const int4 *)(src + 4 * k);
      }
      if (p.dequant) q[g] = qidx[sv * xs + sh];
    }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      if (sv_[g] < 0) continue;
      const int sv = sv_[g], sh = sh_[g];
      int v[CN];
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        if constexpr (S_::narrow) {
          const unsigned w0 = (unsigned)raw[g][k].x, w1 = (unsigned)raw[g][k].y;
          v[4 * k] = vc2_lo16(w0); v[4 * k + 1] = vc2_hi16(w0); v[4 * k + 2] = vc2_lo16(w1); v[4 * k + 3] = vc2_hi16(w1);
        } else { v[4 * k] = raw[g][k].x; v[4 * k + 1] = raw[g][k].y; v[4 * k + 2] = raw[g][k].z; v[4 * k + 3] = raw[g][k].w; }
      }
      if constexpr (S_::narrow) {
        int mn = v[0];
#pragma unroll
        for (int k = 1; k < CN; ++k) mn = min(mn, v[k]);
        if (mn == VC2_ST_SENTINEL) { // values that did not fit 16 bits: the wide plane
          const int32_t *wq = wide + (size_t)(sv * xs + sh) * rs + chunk0;
#pragma unroll
          for (int k = 0; k < CN; ++k) if (v[k] == VC2_ST_SENTINEL) v[k] = wq[k];
        }
      }
      const int i0 = (sv << LH) - ky_base, j0 = (sh << LW) - kx_base;
      const bool cols_in = j0 >= 0 && j0 + BSW <= WXP; // (window origin and width are multiples of four: all or none)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        int *e = v + b * BN;
        if (p.dequant && !VC2_SKIP(p, 8)) {
          const int aq = max(q[g] - qm[b], 0);
          if (aq > 119) atomicOr(p.err, VC2_DEVERR_QINDEX);
          const int qf = qtab[min(aq, 119)], qo = qtab[120 + min(aq, 119)], lim = qtab[240 + min(aq, 119)];
          // scale(), Quantisation.cpp:86-95: (|v| * factor + offset + 2) >> 2 for v != 0 inside the domain (no int
          // overflow: one magnitude test for the band's elements), the literal sequence outside
          unsigned mg[BN], any = 0;
#pragma unroll
          for (int k = 0; k < BN; ++k) { mg[k] = e[k] < 0 ? 0u - (unsigned)e[k] : (unsigned)e[k]; any |= mg[k]; }
          if ((int)any >= 0 && (int)any <= lim) {
#pragma unroll
            for (int k = 0; k < BN; ++k) {
              const unsigned r = mg[k] ? (mg[k] * (unsigned)qf + (unsigned)(qo + 2)) >> 2 : 0u;
              e[k] = e[k] < 0 ? (int)(0u - r) : (int)r;
            }
          } else {
#pragma unroll
            for (int k = 0; k < BN; ++k) e[k] = dequant_f(e[k], qf, qo);
          }
        }
        int *dst = lds + (B0 + b) * C::PLANE + i0 * WXP + j0;
#pragma unroll
        for (int r = 0; r < BSH; ++r) {
          if (i0 + r < 0 || i0 + r >= WYP) continue;
          int *d = dst + r * WXP;
          const int *s = e + r * BSW;
          if (cols_in) {
            if constexpr (BSW == 4) lds_st4(d, {s[0], s[1], s[2], s[3]});
            else if constexpr (BSW == 2) *(int2 *)d = make_int2(s[0], s[1]);
            else d[0] = s[0];
          } else {
#pragma unroll
            for (int c = 0; c < BSW; ++c) if (j0 + c >= 0 && j0 + c < WXP) d[c] = s[c];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// inverse level
// ------------------------------------------------------------------------------------------
// SMALL: the band blocks of a slice are narrower than four coefficients (deep levels); the store bands are then
// gathered slice by slice (see below) instead of window position by window position.
template <int K, bool FINAL, bool SMALL, class ST>
__global__ __launch_bounds__(NTK<K>) void k_inv_fast(const LevelParams p) {
  constexpr int NT = NTK<K>;
  using S_ = St<ST>;
  using C = Cfg<K>;
  extern __shared__ __attribute__((aligned(16))) int lds[];
  const int comp = blockIdx.z % 3, pic = blockIdx.z / 3;
  int tile_x, tile_y;
  if (!tile_of_block(p.tiles_x[comp], p.tiles_y[comp], tile_x, tile_y)) return;
  TILE_STAMP(0);
  constexpr int HY = C::HY, HX = C::HX, WXP = C::WXP, WYP = C::WYP, ACC = WT<K>::accuracy;
  const int out_h = p.in_h[comp], out_w = p.in_w[comp];
  const int y0 = min(tile_y * TY, out_h - TY), x0 = min(tile_x * TX, out_w - TX);
  const int npy = out_h >> 1, npx = out_w >> 1;
  const int ky_base = (y0 - HY) / 2, kx_base = (x0 - HX) / 2;
  const int fh = p.fh[comp], fw = p.fw[comp];
  const int bsh = fh >> 1, bsw = fw >> 1;
  const int lbsh = ilog2(bsh), lbsw = ilog2(bsw);
  const ST *store = (const ST *)p.store + (size_t)pic * p.store_stride;
  const int32_t *wide = S_::narrow ? p.store_wide + (size_t)pic * p.store_stride : nullptr;
  const int32_t *qidx = p.qidx ? p.qidx + (size_t)pic * p.ys * p.xs : nullptr;

  // quant_factor / quant_offset by adjusted index, copied next to the planes: the look-up that follows the
  // slice's index load is then an LDS read instead of a second dependent trip to memory
  // (requested here, put into LDS behind the window's own requests below: the two trips to memory side by side instead of
  // one after the other with a barrier between them -- 1.1 - 1.6 us of a workgroup's 15 - 21 at the deep levels of UHD)
  int *qtab = lds + 4 * C::PLANE;
  int qt_f = 0, qt_o = 0;
  if (threadIdx.x < 120) { qt_f = c_qd.qf[threadIdx.x]; qt_o = c_qd.off[threadIdx.x]; }
  TILE_STAMP(1);

  // ---- gather LL + the three detail bands of tile + halo, dequantising on the way in.
  // Every 16-byte load of the thread (and the slice's quantiser index beside it) is issued before
  // the first one is consumed.
  {
    constexpr int NQI = (WYP * (WXP / 4) + NT - 1) / NT;
    int4 val[4][NQI];             // int16 store / planes: four values in .x, .y
    int qv[4][NQI], kind[4][NQI]; // kind: 0 skip, 1 vector loaded, 2 element-wise path
    const ST *llp = (const ST *)p.ll[comp] + (size_t)pic * p.ll_stride[comp];
    const int32_t *llp_w = S_::narrow ? p.ll_wide[comp] + (size_t)pic * p.ll_stride[comp] : nullptr;
    auto ldq = [](const ST *q) -> int4 { // the raw 4-element load; unpacked (and escapes resolved) when consumed
      if constexpr (S_::narrow) { const uint2 v = *(const uint2 *)q; return make_int4((int)v.x, (int)v.y, 0, 0); }
      else return *(const int4 *)q;
    };
#pragma unroll
    for (int band = 0; band < 4; ++band) {
      const bool from_plane = (band == 0 && !p.ll_from_store);
      const int off = p.coef_off[comp] + (band == 0 ? 0 : p.band_off[comp] * band);
      const bool vec = from_plane ? ((npx & 3) == 0) : (lbsw >= 2);
#pragma unroll
      for (int it = 0; it < NQI; ++it) {
        const int id = it * NT + threadIdx.x;
        kind[band][it] = 0;
        qv[band][it] = 0;
        val[band][it] = make_int4(0, 0, 0, 0);
        if (SMALL && !from_plane) continue; // the store bands come slice by slice, below
        if (id >= WYP * (WXP / 4) || VC2_SKIP(p, 1)) continue;
        const int i = id / (WXP / 4), jq = id - i * (WXP / 4);
        const int by = ky_base + i, bx0 = kx_base + 4 * jq;
        if (by < 0 || by >= npy || bx0 + 4 <= 0 || bx0 >= npx) continue;
        const bool inside = vec && bx0 >= 0 && bx0 + 4 <= npx;
        kind[band][it] = inside ? 1 : 2;
        if (!inside) continue;
        if (from_plane) val[band][it] = ldq(llp + (size_t)by * npx + bx0);
        else {
          const int sv = by >> lbsh, r = by & (bsh - 1), sh = bx0 >> lbsw, c = bx0 & (bsw - 1);
          val[band][it] = ldq(store + (size_t)(sv * p.xs + sh) * p.rec_stride[comp] + off + (r << lbsw) + c);
          if (p.dequant) qv[band][it] = qidx[sv * p.xs + sh];
        }
      }
    }
    if (threadIdx.x < 120) {
      qtab[threadIdx.x] = qt_f; qtab[120 + threadIdx.x] = qt_o;
      // largest magnitude for which |v| * factor + offset + 2 stays below 2^31 (the literal arithmetic otherwise)
      qtab[240 + threadIdx.x] = qt_f > 0 ? (int)((0x7FFFFFFFu - (unsigned)qt_o - 2u) / (unsigned)qt_f) : -1;
    }
    __syncthreads();
#pragma unroll
    for (int band = 0; band < 4; ++band) {
      int *dst = lds + band * C::PLANE;
      const bool from_plane = (band == 0 && !p.ll_from_store);
      const int off = p.coef_off[comp] + (band == 0 ? 0 : p.band_off[comp] * band);
      const int qm = band == 0 ? p.qmatrix[0] : p.qmatrix[p.band + band - 1];
#pragma unroll
      for (int it = 0; it < NQI; ++it) {
        if (kind[band][it] == 0) continue;
        const int id = it * NT + threadIdx.x;
        const int i = id / (WXP / 4), jq = id - i * (WXP / 4);
        const int by = ky_base + i, bx0 = kx_base + 4 * jq;
        int e[4] = {val[band][it].x, val[band][it].y, val[band][it].z, val[band][it].w};
        if constexpr (S_::narrow) {
          if (kind[band][it] == 1) {
            const unsigned w0 = (unsigned)val[band][it].x, w1 = (unsigned)val[band][it].y;
            e[0] = vc2_lo16(w0); e[1] = vc2_hi16(w0); e[2] = vc2_lo16(w1); e[3] = vc2_hi16(w1);
            if (min(min(e[0], e[1]), min(e[2], e[3])) == VC2_ST_SENTINEL) { // values that did not fit 16 bits: the wide plane
              const int32_t *wq;
              if (from_plane) wq = llp_w + (size_t)by * npx + bx0;
              else {
                const int sv = by >> lbsh, r = by & (bsh - 1), sh = bx0 >> lbsw, c = bx0 & (bsw - 1);
                wq = wide + (size_t)(sv * p.xs + sh) * p.rec_stride[comp] + off + (r << lbsw) + c;
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) if (e[k] == VC2_ST_SENTINEL) e[k] = wq[k];
            }
          }
        }
        if (kind[band][it] == 1) {
          if (!from_plane && p.dequant && !VC2_SKIP(p, 8)) {
            const int aq = max(qv[band][it] - qm, 0);
            if (aq > 119) atomicOr(p.err, VC2_DEVERR_QINDEX);
            const int qf = qtab[min(aq, 119)], qo = qtab[120 + min(aq, 119)], lim = qtab[240 + min(aq, 119)];
            // scale(), Quantisation.cpp:86-95: inside the domain (no int overflow) it is (|v| * factor + offset + 2) >> 2
            // for v != 0; one magnitude test for the four values, the literal sequence outside
            unsigned mg[4], any = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { mg[k] = e[k] < 0 ? 0u - (unsigned)e[k] : (unsigned)e[k]; any |= mg[k]; }
            if ((int)any >= 0 && (int)any <= lim) {
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const unsigned r = mg[k] ? (mg[k] * (unsigned)qf + (unsigned)(qo + 2)) >> 2 : 0u;
                e[k] = e[k] < 0 ? (int)(0u - r) : (int)r;
              }
            } else {
#pragma unroll
              for (int k = 0; k < 4; ++k) e[k] = dequant_f(e[k], qf, qo);
            }
          }
        } else if (from_plane) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const size_t at = (size_t)by * npx + min(max(bx0 + k, 0), npx - 1);
            e[k] = S_::load1(llp + at, llp_w + at);
          }
        } else {
          const int sv = by >> lbsh, r = by & (bsh - 1);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int bx = min(max(bx0 + k, 0), npx - 1);
            const int sh = bx >> lbsw, c = bx & (bsw - 1);
            const size_t at = (size_t)(sv * p.xs + sh) * p.rec_stride[comp] + off + (r << lbsw) + c;
            int t = S_::load1(store + at, wide + at);
            if (p.dequant) {
              const int aq = max(qidx[sv * p.xs + sh] - qm, 0);
              if (aq > 119) atomicOr(p.err, VC2_DEVERR_QINDEX);
              t = dequant_f(t, qtab[min(aq, 119)], qtab[120 + min(aq, 119)]);
            }
            e[k] = t;
          }
        }
        lds_st4(dst + i * WXP + 4 * jq, {e[0], e[1], e[2], e[3]});
      }
    }
  }
  TILE_STAMP(2);
  if constexpr (SMALL) {
    // Deep levels: the level's bands of one slice and component are ONE short contiguous run of its record
    // ([LL |] HL | LH | HH, a few to a few dozen coefficients), so they are fetched slice by slice -- one 16-byte load
    // per four coefficients and one index load per slice -- and scattered to their plane positions in LDS, instead of
    // one scattered element load per window position and band.
    const int band_n = p.band_n[comp], lbn = ilog2(band_n);
    const int band_first = p.ll_from_store ? 0 : 1;
    const int chunk0 = p.coef_off[comp] + (p.ll_from_store ? 0 : p.band_off[comp]);
    const int chunk_n = (4 - band_first) * band_n;
    const int sr0 = max(ky_base, 0) >> lbsh, sr1 = min(ky_base + WYP - 1, npy - 1) >> lbsh;
    const int sc0 = max(kx_base, 0) >> lbsw, sc1 = min(kx_base + WXP - 1, npx - 1) >> lbsw;
    const int nsc = sc1 - sc0 + 1, nsl = (sr1 - sr0 + 1) * nsc, nq = (chunk_n + 3) >> 2;
    const bool al = ((chunk0 | p.rec_stride[comp]) & 3) == 0;
    const int qm0 = p.qmatrix[0], qm1 = p.qmatrix[p.band], qm2 = p.qmatrix[p.band + 1], qm3 = p.qmatrix[p.band + 2];
    // Four pieces per thread and turn, all their loads (16 bytes of coefficients, the slice's index) issued before the
    // first is consumed: a turn that loads and consumes one piece at a time waits out a trip to memory per piece -- the
    // phase took 15 us of a workgroup's 22 at the deepest level of UHD, 11 of 17 at the level above (phase stamps).  The
    // two divisions by run-time constants per piece are multiplications (exact for the < 2^16 pieces of a tile).
    constexpr int GB = 4;
    const bool pairs = lbn >= 1 && lbsw >= 1 && (chunk_n & 1) == 0;
    // block shapes known at compile time (every deep level of the BASELINE formats): gather_slices above
    bool shaped = false;
    if (VC2_INV_SHAPED && al && !VC2_SKIP(p, 1)) {
#define VC2_SHAPE(LW_, LN_, LLF_)                                                                                          \
  if (!shaped && lbsw == LW_ && lbn == LN_ && band_first == (LLF_ ? 0 : 1)) {                                              \
    gather_slices<C, NT, ST, LW_, LN_, LLF_>(lds, qtab, p, comp, store, wide, qidx, ky_base, kx_base, npy, npx, chunk0);   \
    shaped = true;                                                                                                         \
  }
      VC2_SHAPE(0, 0, true) VC2_SHAPE(0, 1, true) VC2_SHAPE(1, 1, true) VC2_SHAPE(1, 2, true) VC2_SHAPE(2, 2, true) VC2_SHAPE(2, 3, true)
      VC2_SHAPE(1, 2, false) VC2_SHAPE(2, 2, false) VC2_SHAPE(2, 3, false)
#undef VC2_SHAPE
    }
    const int total = (VC2_SKIP(p, 1) || shaped) ? 0 : nsl * nq;
    const unsigned mg_nq = 0xFFFFFFFFu / (unsigned)nq + 1u, mg_nsc = 0xFFFFFFFFu / (unsigned)nsc + 1u;
    for (int id0 = threadIdx.x; id0 < total; id0 += NT * GB) {
      int e[GB][4], q[GB], sv_[GB], sh_[GB], qd_[GB];
      bool vecl[GB]; // the piece came as one aligned load (else: element by element, when it is consumed)
#pragma unroll
      for (int g = 0; g < GB; ++g) {
        const int id = id0 + g * NT;
        vecl[g] = false; q[g] = 0; sv_[g] = sh_[g] = qd_[g] = 0;
        e[g][0] = e[g][1] = e[g][2] = e[g][3] = 0;
        if (id >= total) continue;
        const int sidx = nq == 1 ? id : (int)__umulhi((unsigned)id, mg_nq), qd = id - sidx * nq; // (the multiplier of 1 does not fit 32 bits)
        const int sr = nsc == 1 ? sidx : (int)__umulhi((unsigned)sidx, mg_nsc), sv = sr0 + sr, sh = sc0 + (sidx - sr * nsc);
        sv_[g] = sv; sh_[g] = sh; qd_[g] = qd;
        const size_t at = (size_t)(sv * p.xs + sh) * p.rec_stride[comp] + chunk0 + 4 * qd;
        if (al && 4 * qd + 4 <= chunk_n) {
          vecl[g] = true;
          if constexpr (S_::narrow) { const uint2 v = *(const uint2 *)(store + at); e[g][0] = (int)v.x; e[g][1] = (int)v.y; }
          else { const int4 v = *(const int4 *)(store + at); e[g][0] = v.x; e[g][1] = v.y; e[g][2] = v.z; e[g][3] = v.w; }
        }
        if (p.dequant) q[g] = qidx[sv * p.xs + sh];
      }
#pragma unroll
      for (int g = 0; g < GB; ++g) {
        const int id = id0 + g * NT;
        if (id >= total) continue;
        const int sv = sv_[g], sh = sh_[g], qd = qd_[g];
        const size_t at = (size_t)(sv * p.xs + sh) * p.rec_stride[comp] + chunk0 + 4 * qd;
        int v4[4];
        if (vecl[g]) {
          if constexpr (S_::narrow) {
            const unsigned w0 = (unsigned)e[g][0], w1 = (unsigned)e[g][1];
            v4[0] = vc2_lo16(w0); v4[1] = vc2_hi16(w0); v4[2] = vc2_lo16(w1); v4[3] = vc2_hi16(w1);
            if (min(min(v4[0], v4[1]), min(v4[2], v4[3])) == VC2_ST_SENTINEL) {
#pragma unroll
              for (int k = 0; k < 4; ++k) if (v4[k] == VC2_ST_SENTINEL) v4[k] = wide[at + k];
            }
          } else { v4[0] = e[g][0]; v4[1] = e[g][1]; v4[2] = e[g][2]; v4[3] = e[g][3]; }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) v4[k] = 4 * qd + k < chunk_n ? S_::load1(store + at + k, wide + at + k) : 0;
        }
        if (pairs) {
          // an aligned pair of the piece lies in one band and one block row (bands and block rows of two or more
          // coefficients): band, row, quantiser constants and the row test once per pair, and scale() as the multiply-add
          // of the vector path above (one magnitude test for the pair, the literal sequence outside its domain)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int idx = 4 * qd + 2 * h;
            if (idx >= chunk_n) continue;
            const int band = band_first + (idx >> lbn), rem = idx & (band_n - 1);
            const int i = (sv << lbsh) + (rem >> lbsw) - ky_base, j = (sh << lbsw) + (rem & (bsw - 1)) - kx_base;
            if (i < 0 || i >= WYP) continue;
            int a = v4[2 * h], b = v4[2 * h + 1];
            if (p.dequant && !VC2_SKIP(p, 8)) {
              const int aq = max(q[g] - (band == 0 ? qm0 : band == 1 ? qm1 : band == 2 ? qm2 : qm3), 0);
              if (aq > 119) atomicOr(p.err, VC2_DEVERR_QINDEX);
              const int qf = qtab[min(aq, 119)], qo = qtab[120 + min(aq, 119)], lim = qtab[240 + min(aq, 119)];
              const unsigned ma = a < 0 ? 0u - (unsigned)a : (unsigned)a, mb = b < 0 ? 0u - (unsigned)b : (unsigned)b;
              if ((int)(ma | mb) >= 0 && (int)(ma | mb) <= lim) {
                const unsigned ra = ma ? (ma * (unsigned)qf + (unsigned)(qo + 2)) >> 2 : 0u, rb = mb ? (mb * (unsigned)qf + (unsigned)(qo + 2)) >> 2 : 0u;
                a = a < 0 ? (int)(0u - ra) : (int)ra;
                b = b < 0 ? (int)(0u - rb) : (int)rb;
              } else { a = dequant_f(a, qf, qo); b = dequant_f(b, qf, qo); }
            }
            int *d = lds + band * C::PLANE + i * WXP + j;
            if (j >= 0 && j < WXP) d[0] = a;
            if (j + 1 >= 0 && j + 1 < WXP) d[1] = b;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int idx = 4 * qd + k;
            if (idx >= chunk_n) continue;
            const int band = band_first + (idx >> lbn), rem = idx & (band_n - 1);
            const int i = (sv << lbsh) + (rem >> lbsw) - ky_base, j = (sh << lbsw) + (rem & (bsw - 1)) - kx_base;
            if (i < 0 || i >= WYP || j < 0 || j >= WXP) continue;
            int v = v4[k];
            if (p.dequant && !VC2_SKIP(p, 8)) {
              const int aq = max(q[g] - (band == 0 ? qm0 : band == 1 ? qm1 : band == 2 ? qm2 : qm3), 0);
              if (aq > 119) atomicOr(p.err, VC2_DEVERR_QINDEX);
              v = dequant_f(v, qtab[min(aq, 119)], qtab[120 + min(aq, 119)]);
            }
            lds[band * C::PLANE + i * WXP + j] = v;
          }
        }
      }
    }
  }
  __syncthreads();
  TILE_STAMP(3);

  // ---- inverse lifting in registers: vertical over every window column, horizontal over the core rows
  {
    constexpr int NQ = WXP / 4;
    constexpr int NITV = (NQ * 2 * (TY / 8) + NT - 1) / NT;
    if (!VC2_SKIP(p, 2)) {
    if constexpr (stepwise<K>()) {
      steps_v<K, true>(lds, 0, NQ, ky_base, npy);
      steps_h<K, true>(lds, HY / 2, TY / 2, kx_base, npx);
      steps_h<K, true>(lds, WYP + HY / 2, TY / 2, kx_base, npx);
    } else {
    v_pass<K, true, NITV>(lds, 0, NQ, ky_base, npy);
    // core rows of both parities: stack rows rp*WYP + HY/2 + [0, TY/2); h_pass takes one contiguous
    // range, so run it once per row parity
    constexpr int NITH = ((TY / 2) * TXQ + NT - 1) / NT;
    h_pass<K, true, NITH>(lds, HY / 2, TY / 2, kx_base, npx);
    h_pass<K, true, NITH>(lds, WYP + HY / 2, TY / 2, kx_base, npx);
    }
    }
  }
  if (VC2_SKIP(p, 4)) return;
  TILE_STAMP(4);

  // ---- interleave, round, write (FINAL: clip + offset + justify + big-endian 16-bit words)
  const int lim_h = FINAL ? p.pic_h[comp] : out_h, lim_w = FINAL ? p.pic_w[comp] : out_w;
  const bool vec_out = FINAL ? (p.word_bytes == 2 && (lim_w & 7) == 0) : ((out_w & 7) == 0);
  for (int id = threadIdx.x; id < TY * (TX / 8); id += NT) {
    const int r = id / TXQ, ch = id % TXQ;
    const int gy = y0 + r, gx0 = x0 + 8 * ch;
    if (gy >= lim_h || gx0 >= lim_w) continue;
    const int *e = lds + ((r & 1) * 2 + 0) * C::PLANE + ((r + HY) >> 1) * WXP + HX / 2 + 4 * ch;
    const I4 a = lds_ld4(e), b = lds_ld4(e + C::PLANE);
    int s[8] = {a.x, b.x, a.y, b.y, a.z, b.z, a.w, b.w};
    if (ACC) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] = (s[k] + (1 << (ACC > 0 ? ACC - 1 : 0))) >> ACC;
    }
    if constexpr (FINAL) {
      uint8_t *row = (uint8_t *)p.plane[comp] + (size_t)pic * p.plane_stride[comp] + (size_t)gy * lim_w * p.word_bytes;
      unsigned u[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = (unsigned)(min(max(s[k], p.clip_lo), p.clip_hi) + p.sample_offset) << p.sample_shift;
      if (vec_out && gx0 + 8 <= lim_w) {
        unsigned wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wv[k] = __builtin_bswap32(((u[2 * k] & 0xFFFFu) << 16) | (u[2 * k + 1] & 0xFFFFu));
        *(uint4 *)(row + (size_t)gx0 * 2) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
      } else {
        for (int k = 0; k < 8; ++k) {
          if (gx0 + k >= lim_w) break;
          uint8_t *q = row + (size_t)(gx0 + k) * p.word_bytes;
          for (int b2 = 0; b2 < p.word_bytes; ++b2) q[b2] = (uint8_t)(u[k] >> (8 * (p.word_bytes - 1 - b2)));
        }
      }
    } else {
      const size_t at = (size_t)pic * p.plane_stride[comp] + (size_t)gy * out_w + gx0;
      ST *row = (ST *)p.plane[comp] + at;
      int32_t *row_w = S_::narrow ? p.plane_wide[comp] + at : nullptr;
      if (vec_out) S_::store8(row, row_w, s);
      else {
#pragma unroll
        for (int k = 0; k < 8; ++k) S_::store1(row + k, row_w + k, s[k]);
      }
    }
  }
  TILE_STAMP(5);
}

template <int K, bool EDGE, bool INV, class ST>
void launch_fast(Launcher &L, const LevelParams &p, int n_pictures, hipStream_t s) {
  int gx = 0, gy = 0;
  for (int c = 0; c < 3; ++c) { gx = std::max(gx, p.tiles_x[c]); gy = std::max(gy, p.tiles_y[c]); }
  dim3 grid(gx, gy, 3 * n_pictures), block(NTK<K>);
  const size_t lds = INV ? Cfg<K>::LDS_INV : Cfg<K>::LDS;
#ifdef VC2HIP_STAMPS
  const char *stamp_file = getenv("VC2HIP_TILE_STAMPS_FILE");
  const size_t stamp_n = (size_t)gx * gy * 3 * n_pictures * 16;
  unsigned long long *d_st = nullptr;
  if (stamp_file) {
    (void)hipMalloc((void **)&d_st, stamp_n * 8);
    (void)hipMemsetAsync(d_st, 0, stamp_n * 8, s);
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tile_stamps), &d_st, sizeof d_st, 0, hipMemcpyHostToDevice, s);
  }
#endif
  if constexpr (INV) {
    // element-wise gather when some component's band blocks are narrower than four coefficients
    bool small = false;
    for (int c = 0; c < 3; ++c) if (p.tiles_x[c] && p.fw[c] / 2 < 4) small = true;
    vc2_prof_begin(L, EDGE ? "idwt_level_final" : "idwt_level", s);
    if (small) {
      vc2_allow_lds((const void *)k_inv_fast<K, EDGE, true, ST>, 160 * 1024);
      VC2_LAUNCH(L, (k_inv_fast<K, EDGE, true, ST>), grid, block, lds, s, p);
    } else {
      vc2_allow_lds((const void *)k_inv_fast<K, EDGE, false, ST>, 160 * 1024);
      VC2_LAUNCH(L, (k_inv_fast<K, EDGE, false, ST>), grid, block, lds, s, p);
    }
  } else {
    vc2_allow_lds((const void *)k_fwd_fast<K, EDGE, ST>, 160 * 1024);
    vc2_prof_begin(L, EDGE ? "dwt_level_first" : "dwt_level", s);
    VC2_LAUNCH(L, (k_fwd_fast<K, EDGE, ST>), grid, block, lds, s, p);
  }
#ifdef VC2HIP_STAMPS
  if (stamp_file) {
    std::vector<unsigned long long> h(stamp_n);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), d_st, stamp_n * 8, hipMemcpyDeviceToHost);
    unsigned long long *none = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stamps), &none, sizeof none);
    (void)hipFree(d_st);
    if (FILE *fp = fopen(stamp_file, "ab")) {
      const int hdr[8] = {INV ? 1 : 0, p.in_w[0], gx, gy, 3 * n_pictures, (int)lds, 0, 0};
      fwrite(hdr, sizeof hdr, 1, fp);
      fwrite(h.data(), 8, stamp_n, fp);
      fclose(fp);
    }
  }
#endif
  vc2_prof_end(L, s);
}

template <bool INV, class ST> int dispatch_fast(Launcher &L, int kernel, bool edge, const LevelParams &p, int n, hipStream_t s) {
#define VC2_CASE(KK)                                                   \
  case KK:                                                             \
    if (edge) launch_fast<KK, true, INV, ST>(L, p, n, s);              \
    else launch_fast<KK, false, INV, ST>(L, p, n, s);                  \
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

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

} // namespace

// The fast kernels apply when every active component has power-of-two slice footprints that divide
// the 32 x 128 tile and a plane at least one tile large.  Rewrites the tiling fields of p.
bool vc2_fast_level_applicable(LevelParams &p) {
  for (int c = 0; c < 3; ++c) {
    if (p.tiles_x[c] == 0 || p.tiles_y[c] == 0) continue;
    if (!pow2(p.fh[c]) || !pow2(p.fw[c]) || p.fh[c] > TY || p.fw[c] > TX || p.fh[c] < 2 || p.fw[c] < 2) return false;
    if (p.in_h[c] < TY || p.in_w[c] < TX || (p.in_w[c] & 7)) return false;
  }
  for (int c = 0; c < 3; ++c) {
    if (p.tiles_x[c] == 0 || p.tiles_y[c] == 0) continue;
    p.tsy[c] = TY / p.fh[c];
    p.tsx[c] = TX / p.fw[c];
    p.tiles_y[c] = (p.in_h[c] + TY - 1) / TY;
    p.tiles_x[c] = (p.in_w[c] + TX - 1) / TX;
  }
  return true;
}
// store16: the store and the level planes hold int16 elements with their wide planes (vc2hip_store.h)
int vc2_launch_forward_fast(Launcher &L, int kernel, bool first, const LevelParams &p, int n, bool store16, hipStream_t s) {
  return store16 ? dispatch_fast<false, int16_t>(L, kernel, first, p, n, s) : dispatch_fast<false, int32_t>(L, kernel, first, p, n, s);
}
int vc2_launch_inverse_fast(Launcher &L, int kernel, bool final_level, const LevelParams &p, int n, bool store16, hipStream_t s) {
  return store16 ? dispatch_fast<true, int16_t>(L, kernel, final_level, p, n, s) : dispatch_fast<true, int32_t>(L, kernel, final_level, p, n, s);
}
