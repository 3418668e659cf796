// The register engine of the streaming transform kernels, shared by vc2hip_dwt_stream.hip (one level per launch) and
// vc2hip_dwt_pair.hip (two consecutive levels per launch): rows of a lane's chunk, horizontal lifting with DPP
// neighbour exchange, the line-based vertical lifting rings, strip / segment geometry of a wavefront.
// (Included inside each translation unit's anonymous namespace: the kernels of the two files are separate instantiations.)
#pragma once

#ifndef VC2_STREAM_MIN_W
#define VC2_STREAM_MIN_W 192 // narrowest plane: a third of the wavefront's lanes at work (below it the tile kernels)
#endif
#ifndef VC2_STREAM_PF
#define VC2_STREAM_PF 2
#endif
constexpr int PF = VC2_STREAM_PF; // row pairs prefetched ahead, forward kernel (divides the ring length RL)
#ifndef VC2_STREAM_PFI
#define VC2_STREAM_PFI 1 // (with the band planes: 1 and 2 the same on the last level, 1 a little ahead below it; 4 slower)
#endif
constexpr int PFI = VC2_STREAM_PFI; // the same for the inverse kernel (four loads per pair and lane)

// one row of a lane's chunk: NP coefficient pairs -- elements 0..NP-1 the even columns, NP..2NP-1 the odd columns.
// NP = 4: a chunk of 8 samples (the level a wavefront reads from memory); NP = 2: the four LL samples that level leaves
// per lane, i.e. the chunk of the NEXT level when two levels run in one kernel (vc2hip_dwt_pair.hip)
template <int NP> struct RowT {
  int v[2 * NP];
  __device__ __forceinline__ int &operator[](int i) { return v[i]; }
  __device__ __forceinline__ const int &operator[](int i) const { return v[i]; }
};
using Row = RowT<4>;
template <int NP> __device__ __forceinline__ RowT<NP> operator+(const RowT<NP> &a, const RowT<NP> &b) { RowT<NP> r; _Pragma("unroll") for (int i = 0; i < 2 * NP; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
template <int NP> __device__ __forceinline__ RowT<NP> operator-(const RowT<NP> &a, const RowT<NP> &b) { RowT<NP> r; _Pragma("unroll") for (int i = 0; i < 2 * NP; ++i) r.v[i] = a.v[i] - b.v[i]; return r; }
template <int NP> __device__ __forceinline__ RowT<NP> operator-(const RowT<NP> &a) { RowT<NP> r; _Pragma("unroll") for (int i = 0; i < 2 * NP; ++i) r.v[i] = -a.v[i]; return r; }
template <int NP> __device__ __forceinline__ RowT<NP> operator+(const RowT<NP> &a, int b) { RowT<NP> r; _Pragma("unroll") for (int i = 0; i < 2 * NP; ++i) r.v[i] = a.v[i] + b; return r; }
template <int NP> __device__ __forceinline__ RowT<NP> operator*(int b, const RowT<NP> &a) { RowT<NP> r; _Pragma("unroll") for (int i = 0; i < 2 * NP; ++i) r.v[i] = b * a.v[i]; return r; }
template <int NP> __device__ __forceinline__ RowT<NP> operator>>(const RowT<NP> &a, int b) { RowT<NP> r; _Pragma("unroll") for (int i = 0; i < 2 * NP; ++i) r.v[i] = a.v[i] >> b; return r; }
template <int NP> __device__ __forceinline__ RowT<NP> vc2_times9(const RowT<NP> &a) { RowT<NP> r; _Pragma("unroll") for (int i = 0; i < 2 * NP; ++i) r.v[i] = ::vc2_times9(a.v[i]); return r; }

template <int CTRL> __device__ __forceinline__ int dppm(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false);
}
// streaming stores: what these kernels write is read by the NEXT kernel, gigabytes later -- nothing in L2 is worth displacing
// for it (round 5: the decoded picture's rows written this way take the last inverse level from a placement-dependent
// 1.75 ms per 128 UHD pictures in most processes to 1.34 - 1.41 in most)
typedef int vc2_nt4_t __attribute__((ext_vector_type(4)));
typedef int vc2_nt2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_nt(void *p, uint4 v) {
  const vc2_nt4_t x = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
  __builtin_nontemporal_store(x, (__attribute__((address_space(1))) vc2_nt4_t *)(size_t)p);
}
__device__ __forceinline__ void st_nt(void *p, uint2 v) {
  const vc2_nt2_t x = {(int)v.x, (int)v.y};
  __builtin_nontemporal_store(x, (__attribute__((address_space(1))) vc2_nt2_t *)(size_t)p);
}
__device__ __forceinline__ void st_nt(void *p, unsigned v) { __builtin_nontemporal_store((int)v, (__attribute__((address_space(1))) int *)(size_t)p); }

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------
// lifting step tables of wavelet K as constexpr functions of the step number
// ------------------------------------------------------------------------------------------
template <int K> constexpr int kdmin(int s) {
  return s == 0 ? step_dmin<K, 0>() : s == 1 ? step_dmin<K, 1>() : s == 2 ? step_dmin<K, 2>() : step_dmin<K, 3>();
}
template <int K> constexpr int kdmax(int s) {
  return s == 0 ? step_dmax<K, 0>() : s == 1 ? step_dmax<K, 1>() : s == 2 ? step_dmax<K, 2>() : step_dmax<K, 3>();
}
template <int K> constexpr bool kodd(int s) {
  return s == 0 ? step_targets_odd<K, 0>() : s == 1 ? step_targets_odd<K, 1>() : s == 2 ? step_targets_odd<K, 2>() : step_targets_odd<K, 3>();
}
// halo lanes per strip side: the lifting steps of a row reach sum(|dmin|) pairs to the left and sum(dmax) to the right
template <int K, int NP = 4> constexpr int halo_lanes() {
  int l = 0, r = 0;
  for (int s = 0; s < WT<K>::nsteps; ++s) { l -= kdmin<K>(s); r += kdmax<K>(s); }
  return ((l > r ? l : r) + NP - 1) / NP;
}
// a lifting step's taps beyond the lane's chunk come from the ADJACENT lane only
template <int K, int NP> constexpr bool reach_fits() {
  for (int s = 0; s < WT<K>::nsteps; ++s) if (-kdmin<K>(s) > NP || kdmax<K>(s) > NP) return false;
  return true;
}

// ------------------------------------------------------------------------------------------
// horizontal lifting of one row in registers
// ------------------------------------------------------------------------------------------
// redge: this lane holds the plane's last chunk, but is not the wavefront's last lane (planes narrower than a wavefront):
// its taps beyond the chunk clamp to its own last pair like the last lane's do
template <int K, int S, bool INV, int NP> __device__ __forceinline__ void h_step(RowT<NP> &r, bool redge) {
  constexpr bool odd = step_targets_odd<K, S>();
  constexpr int dmin = step_dmin<K, S>(), dmax = step_dmax<K, S>();
  static_assert(-dmin <= NP && dmax <= NP, "taps beyond the adjacent lane");
  constexpr int TO = odd ? NP : 0, UO = odd ? 0 : NP; // target / source parity inside the row
  int W[NP + dmax - dmin];
#pragma unroll
  for (int j = dmin; j <= NP - 1 + dmax; ++j) {
    if (j < 0) W[j - dmin] = dppm<0x138>(r[UO], r[UO + NP + j]);          // left neighbour (wave_shr:1)
    else if (j > NP - 1) {                                                // right neighbour (wave_shl:1)
      const int nb = dppm<0x130>(r[UO + NP - 1], r[UO + j - NP]);
      W[j - dmin] = redge ? r[UO + NP - 1] : nb;
    } else W[j - dmin] = r[UO + j];
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int d = lift_delta<K, S>([&](int t) -> int { return W[i + t - dmin]; });
    r[TO + i] = INV ? r[TO + i] - d : r[TO + i] + d;
  }
}
template <int K, bool INV, int NP> __device__ __forceinline__ void h_lift(RowT<NP> &r, bool redge) {
  constexpr int N = WT<K>::nsteps;
  if constexpr (!INV) {
    h_step<K, 0, false, NP>(r, redge);
    h_step<K, 1, false, NP>(r, redge);
    if constexpr (N == 4) { h_step<K, 2, false, NP>(r, redge); h_step<K, 3, false, NP>(r, redge); }
  } else {
    if constexpr (N == 4) { h_step<K, 3, true, NP>(r, redge); h_step<K, 2, true, NP>(r, redge); }
    h_step<K, 1, true, NP>(r, redge);
    h_step<K, 0, true, NP>(r, redge);
  }
}

// ------------------------------------------------------------------------------------------
// vertical lifting, line based.  Positions p = 0..N-1 are the lifting steps in processing order (forward: step p,
// inverse: step N-1-p).  With row pairs up to index m loaded, position p is complete up to c_p = m - OFF_p,
// OFF_p = dmax_0 + ... + dmax_p.  Sequences: the raw rows of either parity (complete to m) and X_p, the output of
// position p (complete to c_p).  Position p reads X_{p-1} (raw for p = 0) at c_p + dmin_p .. c_p + dmax_p and
// updates X_{p-2} (raw for p < 2) at c_p.  A sequence keeps the rows its consumers still need.
// ------------------------------------------------------------------------------------------
template <int K, bool INV> struct VT {
  static constexpr int N = WT<K>::nsteps;
  static constexpr int sid(int p) { return INV ? N - 1 - p : p; }
  static constexpr int dmin(int p) { return kdmin<K>(sid(p)); }
  static constexpr int dmax(int p) { return kdmax<K>(sid(p)); }
  static constexpr bool odd(int p) { return kodd<K>(sid(p)); }
  static constexpr int off(int p) { int s = 0; for (int t = 0; t <= p; ++t) s += dmax(t); return s; }
  static constexpr int sum_dmin() { int s = 0; for (int t = 0; t < N; ++t) s += dmin(t); return s; }
  static constexpr int OFFL = off(N - 1);
  static constexpr int mx(int a, int b) { return a > b ? a : b; }
  static constexpr int len_raw(bool parity_odd) {
    if (parity_odd == odd(0)) return off(0) + 1;             // updated by position 0
    return mx(dmax(0) - dmin(0), N >= 2 ? off(1) : 0) + 1;   // read by position 0, updated by position 1
  }
  static constexpr int len_x(int p) {
    int d = 0;
    if (p + 1 < N) d = mx(d, dmax(p + 1) - dmin(p + 1));
    if (p + 2 < N) d = mx(d, dmax(p + 1) + dmax(p + 2));
    if (p >= N - 2) d = mx(d, off(N - 1) - off(p));
    return d + 1;
  }
};

// Every sequence lives in a ring of RL register rows: the row of index i sits in slot i mod RL.  The walk is unrolled
// RL times (phase U = (m - m0) mod RL is a compile-time constant inside each copy), so every slot number is a constant
// and no row is ever moved; a new row overwrites the one RL indices older, which no consumer needs any more
// (the window lengths of VT are at most 4 for every wavelet but Fidelity, whose rings are 8 long).
template <int K> constexpr int RLK = K == VC2HIP_FIDELITY ? 8 : 4; // (Fidelity: windows of 8 rows; the slots of a ring that no consumer reads any more are dead registers)
template <int K, bool INV, int NP = 4> struct VEng {
  using T = VT<K, INV>;
  using Row = RowT<NP>;
  static constexpr int RL = RLK<K>;
  static_assert(T::len_raw(false) <= RL && T::len_raw(true) <= RL && T::len_x(0) <= RL && T::len_x(1) <= RL &&
                (T::N < 3 || (T::len_x(2) <= RL && T::len_x(3) <= RL)), "row ring too short for this wavelet");
  Row rw[2][RL]; // raw rows: [0] even rows, [1] odd rows
  Row x[4][RL];  // outputs of the positions
  static constexpr int sl(int rel) { return ((rel % RL) + RL) % RL; } // slot of the row `rel` indices from row m, at phase 0

  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int j = 0; j < RL; ++j)
#pragma unroll
      for (int k = 0; k < 2 * NP; ++k) { rw[0][j].v[k] = 0; rw[1][j].v[k] = 0; x[0][j].v[k] = 0; x[1][j].v[k] = 0; x[2][j].v[k] = 0; x[3][j].v[k] = 0; }
  }
  // put the row of index `first ? 0 : any` into its slot; index 0 also fills the ring (the rows above the plane replicate it)
  template <int SLOT> static __device__ __forceinline__ void put(Row (&w)[RL], const Row &r, bool first) {
    // (element by element, and no store common to two paths: a merged store or copy would address the ring through a
    // variable and force all of it into scratch memory)
#pragma unroll
    for (int k = 0; k < 2 * NP; ++k) w[SLOT].v[k] = r.v[k];
    if (first) {
#pragma unroll
      for (int j = 0; j < RL; ++j)
        if (j != SLOT) {
#pragma unroll
          for (int k = 0; k < 2 * NP; ++k) w[j].v[k] = r.v[k];
        }
    }
  }
  template <int SRC, int DST> static __device__ __forceinline__ void copy(Row (&w)[RL]) {
#pragma unroll
    for (int k = 0; k < 2 * NP; ++k) w[DST].v[k] = w[SRC].v[k];
  }
  // The steady state carries no edge handling: a conditional fill of a ring would be turned into selects that read every
  // slot in every iteration and keep all of them alive.
  // Walks start at multiples of RL, so the phase is m mod RL everywhere and every edge decision is known at compile time
  // (the bottom of a plane whose pair count is not a multiple of RL: one tail per remainder, see the kernels):
  // MODE 0: steady state.  MODE 1: the first RL iterations of a walk that starts at the plane's top (a sequence
  // receives its index 0 in iteration off(P)).  MODE 2: the OFFL iterations below the plane's last pair (m = np + D:
  // nothing is loaded; position P repeats its last row once U + 1 > off(P)).
  template <int U, int P, int MODE, int D> __device__ __forceinline__ void pos(int m, int np) {
    if constexpr (P < T::N) {
      constexpr int dst = sl(U - T::off(P));
      constexpr int dmin = T::dmin(P), dmax = T::dmax(P);
      if constexpr (MODE == 2 && (D + 1 > T::off(P))) copy<sl(U - T::off(P) - 1), dst>(x[P]); // below the plane: the last pair again
      else {
        Row W[dmax - dmin + 1]; // rows c_P + dmin .. c_P + dmax of what position P reads
#pragma unroll
        for (int t = dmin; t <= dmax; ++t) {
          if constexpr (P == 0) W[t - dmin] = rw[T::odd(0) ? 0 : 1][sl(U - T::off(P) + t)];
          else W[t - dmin] = x[P - 1][sl(U - T::off(P) + t)];
        }
        const Row d = lift_delta<K, T::sid(P)>([&](int t) -> Row { return W[t - dmin]; });
        Row o;
        if constexpr (P < 2) o = rw[T::odd(P) ? 1 : 0][dst];
        else o = x[P - 2][dst];
        put<dst>(x[P], INV ? o - d : o + d, MODE == 1 && U == T::off(P));
      }
      pos<U, P + 1, MODE, D>(m, np);
    }
  }
  // row pair m (even row re, odd row ro; ignored below the plane) enters at phase U; afterwards pair m - OFFL is complete.
  // MODE 2: m = np + D, D = 0 .. OFFL-1 (the phase is (np + D) mod RL: planes need not hold a multiple of RL pairs)
  template <int U, int MODE, int D = U> __device__ __forceinline__ void step(int m, int np, const Row &re, const Row &ro) {
    if constexpr (MODE == 2) { copy<sl(U - 1), U>(rw[0]); copy<sl(U - 1), U>(rw[1]); }
    else { put<U>(rw[0], re, MODE == 1 && U == 0); put<U>(rw[1], ro, MODE == 1 && U == 0); }
    pos<U, 0, MODE, D>(m, np);
  }
  // Parking (vc2hip_dwt_pair.hip, level b of the inverse kernel): between two steps only the rows inside the sequences'
  // windows are alive -- len_raw / len_x rows back from each sequence's newest.  park<U>() after the step of phase U
  // writes exactly those slots to an LDS area of park_rows() rows (row r of lane l at (r * 64 + l) * 2 NP dwords),
  // unpark<U>() before the next step reads them back: in between the engine holds no register.
  static constexpr int park_rows() { int n = T::len_raw(false) + T::len_raw(true); for (int p = 0; p < T::N; ++p) n += T::len_x(p); return n; }
  template <int U, bool STORE> __device__ __forceinline__ void park_io(int *area, int lane) {
    int r = 0;
    auto io = [&](Row &row) __attribute__((always_inline)) {
      int *q = area + (r * 64 + lane) * 2 * NP;
      ++r;
      static_assert(NP == 2 || NP == 4, "16- or 32-byte rows");
      if constexpr (STORE) {
        *(int4 *)q = make_int4(row.v[0], row.v[1], row.v[2], row.v[3]);
        if constexpr (NP == 4) *(int4 *)(q + 4) = make_int4(row.v[4], row.v[5], row.v[6], row.v[7]);
      } else {
        const int4 a = *(const int4 *)q;
        row.v[0] = a.x; row.v[1] = a.y; row.v[2] = a.z; row.v[3] = a.w;
        if constexpr (NP == 4) { const int4 b = *(const int4 *)(q + 4); row.v[4] = b.x; row.v[5] = b.y; row.v[6] = b.z; row.v[7] = b.w; }
      }
    };
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int j = 0; j < RL; ++j) if (j < T::len_raw(par == 1)) io(rw[par][sl(U - j)]);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int j = 0; j < RL; ++j) if (p < T::N && j < T::len_x(p < T::N ? p : 0)) io(x[p][sl(U - T::off(p < T::N ? p : 0) - j)]);
  }
  template <int U> __device__ __forceinline__ void park(int *area, int lane) { park_io<U, true>(area, lane); }
  template <int U> __device__ __forceinline__ void unpark(int *area, int lane) { park_io<U, false>(area, lane); }
  // the completed pair m - OFFL: its even / odd row in the latest version
  template <int U> __device__ __forceinline__ const Row &out(bool odd_row) const {
    constexpr int N = T::N, s = sl(U - T::OFFL);
    if (T::odd(N - 1) == odd_row) return x[N - 1][s];
    return x[N - 2][s];
  }
};

// ------------------------------------------------------------------------------------------
// strip / segment geometry of one wavefront
// ------------------------------------------------------------------------------------------
struct Strip {
  int c0;           // first chunk of the wavefront
  int lo, hi;       // lanes [lo, hi) own their results
  int nsl, sx0;     // slices across the owned lanes, first slice
  int kA, kB;       // output row pairs [kA, kB)
};
// Work items are (picture, component, strip, segment); all of them cost the same.  The grid is one-dimensional and holds
// working items only, numbered so that the eight XCDs (consecutive workgroups go to consecutive XCDs: block b runs on
// XCD b mod 8) get equal shares and an XCD walks down the segments of one strip -- neighbouring segments share their
// run-in rows in that XCD's L2: b = (group * segments + segment) * 8 + column mod 8, column = group * 8 + b mod 8, the
// columns being all (picture, strip) of the luma planes, then of the U planes, then of the V planes.
// (Round 2 launched a strips x segments x (3 * pictures) grid: with 8 luma and 4 chroma strips the XCD was the strip
// number, XCDs 0-3 got three times the work of XCDs 4-7 and half the chip idled for half of the kernel.)
// A workgroup is VC2_STREAM_WG_WAVES wavefronts, each with a work item of its own and no barrier between them: 1 (one
// wavefront per workgroup, rounds 2 - 4), or 4 -- the four wavefronts of a workgroup go one to each SIMD of a CU whatever
// the dispatcher's state.  Round 4 built the second to see whether wavefront placement explains why the level-0
// transforms run at 0.45 ms per 32 UHD pictures in some processes and at 0.50 in others (tools/probe/bimodal*.py): it
// does not -- on a box that shows both modes both workgroup shapes show them, a second box is always fast, a third always
// slow (A/B of the two builds, five processes each per box).  Neither do the buffers' addresses, their size rounding, the
// stream or the workgroup-to-XCD rotation (tools/probe/xcd_map.hip, stream_bw.hip): the mode follows the box and the
// moment, i.e. the GPU's clock / power state.  Item of wavefront w of workgroup g: 8 * (4 * (g / 8) + w) + g % 8 -- item
// mod 8 is still the workgroup's XCD.
#ifndef VC2_STREAM_WG_WAVES
#define VC2_STREAM_WG_WAVES 1
#endif
__device__ __forceinline__ int stream_wave() { return VC2_STREAM_WG_WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0; }
__device__ __forceinline__ int stream_item() {
  if (VC2_STREAM_WG_WAVES == 1) return (int)blockIdx.x;
  const int g = (int)blockIdx.x;
  return 8 * (VC2_STREAM_WG_WAVES * (g >> 3) + stream_wave()) + (g & 7);
}
template <int HLN> __device__ __forceinline__ bool strip_of_block_h(const LevelParams &p, int &comp, int &pic, Strip &s) {
  const int b = stream_item(), x = b & 7, t = b >> 3;
  const int seg = t % p.st_segmax, col = (t / p.st_segmax) * 8 + x;
  const int n0 = p.st_npic * p.st_strips[0], n1 = p.st_npic * p.st_strips[1], n2 = p.st_npic * p.st_strips[2];
  int within;
  if (col < n0) { comp = 0; within = col; }
  else if (col < n0 + n1) { comp = 1; within = col - n0; }
  else if (col < n0 + n1 + n2) { comp = 2; within = col - n0 - n1; }
  else return false;
  if (seg >= p.st_segs[comp]) return false; // (components of different heights: 4:2:0)
  pic = within / p.st_strips[comp];
  const int strip = within - pic * p.st_strips[comp];
  const int nch = p.in_w[comp] >> 3, out = p.st_out[comp];
  s.c0 = max(min(max(strip * out - HLN, 0), nch - 64), 0); // (a plane narrower than 64 chunks: one strip from chunk 0, idle lanes behind it)
  s.lo = strip * out - s.c0;
  s.hi = min((strip + 1) * out, nch) - s.c0;
  s.nsl = (s.hi - s.lo) >> p.st_llps[comp];
  s.sx0 = (strip * out) >> p.st_llps[comp];
  const int bsh = p.fh[comp] >> 1; // segment = the slice rows [seg * ys / nseg, (seg + 1) * ys / nseg)
  s.kA = (int)(__umul24(seg, p.ys) / (unsigned)p.st_segmax) * bsh;
  s.kB = (int)(__umul24(seg + 1, p.ys) / (unsigned)p.st_segmax) * bsh;
  return true;
}

// Issue priority in turn.  A SIMD serves its oldest wavefront first; with every wavefront slot filled by one launch the
// first-dispatched wavefront of a SIMD finished in 150 us and the last-dispatched in 260 (measured, inverse level 0), and
// the chip idled through that tail.  Wavefronts that take turns at the highest priority (a new turn every ring block) run
// at the same pace and end together.
__device__ __forceinline__ void prio_turn(int v) {
  switch (v & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}
__device__ __forceinline__ int ilog2d(int v) { return 31 - __clz(v); }
// address arithmetic: row / slice numbers and row / record lengths are far below 2^24 and their products (element offsets
// inside one picture) below 2^32, so the full-rate 24-bit multiply serves instead of 64-bit multiplies
__device__ __forceinline__ size_t mul24z(int a, int b) { return (size_t)__umul24((unsigned)a, (unsigned)b); }

template <int K> __device__ __forceinline__ bool strip_of_block(const LevelParams &p, int &comp, int &pic, Strip &s) {
  return strip_of_block_h<halo_lanes<K>()>(p, comp, pic, s);
}

