// Two consecutive transform levels in ONE launch (round 5).
//
// The one-level streaming kernels (vc2hip_dwt_stream.hip) write every LL row to a plane that the next launch reads back,
// and every level pays its own launch, its own filter run-in per segment and -- for the levels whose slice footprint is
// narrower than a lane's chunk -- the LDS tile kernels (levels 2 and 3 of UHD 4:2:2 took seven times their share of
// the samples).  The reference's level loop (WaveletTransform.cpp:262-281 forward, :321-342 inverse) forces none of
// that: level l+1 only ever needs the LL rows of level l in order.  Here a wavefront that walks down level a's rows
// (a lane = 8 samples = 4 coefficient pairs per row, VEng<K, INV, 4>) keeps the 4 LL samples a row pair leaves per lane
// in registers and feeds them, two rows at a time, to a second ring engine over half rows (VEng<K, INV, 2>: the same
// lanes, 2 pairs per lane, the same DPP neighbour exchange; strips overlap by the halo of both levels):
//   forward: level a's LL plane is never written -- level b's bands and LL leave the kernel;
//   inverse: level b's output rows are level a's LL rows -- the plane between them is never read.
// Level a's unrolled walk has eight phases (its own four ring slots times the two LL rows of a level-b pair), so every
// ring slot of both engines is a compile-time constant, as in the one-level kernels.
// Bands reach the coefficient store through wavefront-private LDS images in RECORD order, [slice][the level's contiguous
// run of a component record], flushed as 16-byte pieces (8 / 4 bytes where a run is shorter): with the run as the unit
// a lane may hold one, a part of one, or TWO slices' columns (slice footprints of 4 samples: the deep levels of 4:2:2
// chroma, which the one-level kernels refuse), and the last level's [LL | HL | LH | HH] is one run.
// Same results as the one-level kernels and the tile kernels, which remain for everything this file does not take
// (Fidelity's and Daub97's long filters, planes whose height is not a multiple of four rows, one segment per strip on
// planes whose pair count is not a multiple of eight, ...): see vc2_pair_applicable.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

#include "vc2hip_internal.h"
#include "vc2hip_store.h"
#include "vc2hip_wavelets.h"

void vc2_prof_begin(Launcher &L, const char *name, hipStream_t s);
void vc2_prof_end(Launcher &L, hipStream_t s);

__constant__ QuantTables c_qsp;
void vc2_upload_tables_pair(const QuantTables &t, hipStream_t s) {
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(c_qsp), &t, sizeof t, 0, hipMemcpyHostToDevice, s);
}

namespace {

#ifndef VC2_PAIR_NT
#define VC2_PAIR_NT 1 // the slice records leave k_fwd_pair with non-temporal stores (the slice coder reads them gigabytes later)
#endif
#include "vc2hip_stream_eng.h"

// Issue priority in turn (prio_turn, vc2hip_stream_eng.h), for kernels with TWO wavefronts per SIMD.  The four-level
// rotation of the one-level kernels gives the second wavefront of a SIMD the higher priority in three turns of four
// ((t + 1) & 3 > t & 3); mode: 1 = that rotation, 2 = strict alternation (3 / 0), a turn = one block of eight row pairs;
// >= 3: alternation with turns of 2^(mode - 2) blocks
__device__ __forceinline__ void pair_prio(int mode, int prio0, int block) {
  if (mode == 1) { prio_turn(prio0 + block); return; }
  const int t = mode == 2 ? block : block >> (mode - 2);
  if ((prio0 + t) & 1) __builtin_amdgcn_s_setprio(3);
  else __builtin_amdgcn_s_setprio(0);
}

template <int V> using IC = std::integral_constant<int, V>;
template <bool V> using BC = std::integral_constant<bool, V>;

// wavelets whose two lifting steps reach no further than the adjacent lane at two pairs per lane
template <int K> constexpr bool pair_kernel() {
  if constexpr (K == VC2HIP_FIDELITY || K == VC2HIP_DAUB97) return false;
  else return reach_fits<K, 2>() && reach_fits<K, 4>() && WT<K>::nsteps == 2;
}
template <int K> constexpr int pair_halo() { return halo_lanes<K, 4>() + halo_lanes<K, 2>(); }
constexpr int fdiv2(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); } // floor(v / 2)

// ------------------------------------------------------------------------------------------
// forward: levels a and b = a + 1
// ------------------------------------------------------------------------------------------
// wavefronts per SIMD the register allocator leaves room for: this kernel's time is its instruction stream at the rate ONE
// wavefront issues (about an instruction per 8 cycles and wavefront, DESIGN.md "What a SIMD issues"), so the third wavefront
// per SIMD (168 registers) is worth a third of the time; DD137's longer windows need the 256 of two
#ifndef VC2_PAIR_WPE
#define VC2_PAIR_WPE 2
#endif
template <int K> constexpr int pair_wpe() { return K == VC2HIP_DD137 ? 2 : VC2_PAIR_WPE; }
#ifndef VC2_PAIR_PF
#define VC2_PAIR_PF 2
#endif
constexpr int PFP = VC2_PAIR_PF; // row pairs of level a prefetched ahead
template <int K, bool FIRST, class ST>
__global__ __launch_bounds__(64, (pair_wpe<K>())) void k_fwd_pair(const PairParams pp) {
  using S_ = St<ST>;
  using EA = VEng<K, false, 4>;
  using EB = VEng<K, false, 2>;
  using T = VT<K, false>;
  constexpr int OFFL = T::OFFL, SD = T::sum_dmin();
  constexpr int ACC = WT<K>::accuracy;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#define p pp.a  /* (the kernel argument's members by name: a reference to a member of the argument sends all of it to scratch memory) */
#define pb pp.b
  const int lane = threadIdx.x & 63;
  int comp, pic;
  Strip sp;
  if (!strip_of_block_h<pair_halo<K>()>(p, comp, pic, sp)) return;
  const int spl = pp.spl[comp]; // slices per lane (wave-uniform): 1, or 2 for footprints of 4 samples (then one lane per "slice group")
  sp.nsl *= spl;
  sp.sx0 *= spl;
  // LDS of the wavefront: grp_a images of level a (one per slice row of a group), grp_a + 1 of level b (a ring: level b's
  // slice row completes 2 * OFFL pairs behind level a's), a ring of LL rows of level b -- everything leaves in ONE burst
  // of stores per group (see `burst`)
  ST *imgA = (ST *)smem, *imgB = (ST *)(smem + pp.img_b);
  unsigned *ring_ll = (unsigned *)(smem + pp.ring_ll); // per row 64 lanes x (one dword: two 16-bit elements; two dwords: two int32)
  const int szA = pp.sz_a, szB = pp.sz_b, grpA = pp.grp_a, nll = pp.n_ll; // image sizes in elements; slice rows per burst; LL ring rows
  // every member of the argument the walk uses, read ONCE here: the unrolled walk below would otherwise hold hundreds of
  // uses of the argument, and beyond 300 the compiler stops reading a by-value argument in place and copies all of it to
  // scratch memory (instcombine-max-copied-from-constant-users)
  const int xs = p.xs, slice_coefs = p.slice_coefs, st_prio = p.st_prio;
#ifdef VC2HIP_ABLATE // timing experiments (VC2HIP_DEBUG_SKIP): 8 no level-b lifting, 16 level b's results stay in the wavefront, 32 level a's
  const int dskip = p.debug_skip;
#else
  constexpr int dskip = 0;
#endif
  const int pieceA = pp.piece_a[comp], pieceB = pp.piece_b[comp];
  const int sshift = FIRST ? (comp ? p.sample_shift_c : p.sample_shift) : 0;
  const int bias = FIRST ? -((comp ? p.sample_offset_c : p.sample_offset) << ACC) : 0;
  const int in_h = p.in_h[comp], in_w = p.in_w[comp], npA = in_h >> 1, npB = npA >> 1;
  const int chunk = min(sp.c0 + lane, (in_w >> 3) - 1);
  const bool own = lane >= sp.lo && lane < sp.hi;
  const bool redge = sp.c0 + lane == (in_w >> 3) - 1 && lane != 63;

  // ---- input rows of level a (as k_fwd_stream)
  const uint8_t *raw = nullptr;
  const ST *lvl = nullptr;
  const int32_t *lvl_w = nullptr;
  if constexpr (FIRST) raw = (const uint8_t *)p.plane[comp] + (size_t)pic * p.plane_stride[comp] + (size_t)chunk * 16;
  else {
    lvl = (const ST *)p.plane[comp] + (size_t)pic * p.plane_stride[comp] + (size_t)chunk * 8;
    if constexpr (S_::narrow) lvl_w = p.plane_wide[comp] + (size_t)pic * p.plane_stride[comp] + (size_t)chunk * 8;
  }
  const int pic_h = FIRST ? p.pic_h[comp] : in_h;
  constexpr int NQ = (FIRST || S_::narrow) ? 1 : 2;
  uint4 pf[PFP][2][NQ];
  auto fetch = [&](int m, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int y = min(2 * m + h, pic_h - 1); // waveletPad: rows below the picture replicate its last row
      if constexpr (FIRST) pf[slot][h][0] = *(const uint4 *)(raw + mul24z(y, in_w) * 2);
      else {
        const ST *q = lvl + mul24z(y, in_w);
        pf[slot][h][0] = *(const uint4 *)q;
        if constexpr (NQ == 2) pf[slot][h][1] = *(const uint4 *)(q + 4);
      }
    }
  };
  auto convert = [&](int m, int slot, int h, RowT<4> &r) __attribute__((always_inline)) {
    if constexpr (FIRST) {
      const unsigned w[4] = {pf[slot][h][0].x, pf[slot][h][0].y, pf[slot][h][0].z, pf[slot][h][0].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned b = __builtin_amdgcn_perm(w[k], w[k], 0x02030001u); // both big-endian 16-bit words to host order
        r[k] = (int)(((b & 0xFFFFu) >> sshift) << ACC) + bias;
        r[4 + k] = (int)((b >> (16 + sshift)) << ACC) + bias;
      }
    } else {
      int s[8];
      if constexpr (S_::narrow) S_::unpack8(pf[slot][h][0], lvl_w + mul24z(2 * m + h, in_w), s);
      else {
        s[0] = (int)pf[slot][h][0].x; s[1] = (int)pf[slot][h][0].y; s[2] = (int)pf[slot][h][0].z; s[3] = (int)pf[slot][h][0].w;
        s[4] = (int)pf[slot][h][NQ - 1].x; s[5] = (int)pf[slot][h][NQ - 1].y; s[6] = (int)pf[slot][h][NQ - 1].z; s[7] = (int)pf[slot][h][NQ - 1].w;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { r[k] = (int)((unsigned)s[2 * k] << ACC); r[4 + k] = (int)((unsigned)s[2 * k + 1] << ACC); }
    }
  };

  // ---- outputs.  A component record holds [LL | bands of the coarsest level | ... | bands of the finest]; the bands
  // HL | LH | HH of one level are one contiguous run of 3 * bn coefficients (with LL in front of it at the last level)
  const int bshA = p.fh[comp] >> 1, bswA = p.fw[comp] >> 1, lbshA = ilog2d(bshA), bnA = bshA * bswA;
  const int bshB = bshA >> 1, bswB = bswA >> 1, lbshB = lbshA - 1, bnB = bshB * bswB;
  const int b_last = pb.ll_to_store;
  const int run0A = p.coef_off[comp] + bnA, run0B = pb.coef_off[comp] + (b_last ? 0 : bnB);
  const int ssA = pp.ss_a[comp], ssB = pp.ss_b[comp];
  const int llps = p.st_llps[comp];
  const int si = spl == 1 ? (lane - sp.lo) >> llps : (lane - sp.lo) * spl; // the lane's (first) slice inside the strip
  const int cl = spl == 1 ? (lane - sp.lo) & ((1 << llps) - 1) : 0;        // the lane's chunk inside its slice
  ST *store = (ST *)p.store + (size_t)pic * p.store_stride;
  int32_t *wide = S_::narrow ? p.store_wide + (size_t)pic * p.store_stride : nullptr;
  ST *llp = nullptr;
  int32_t *llp_w = nullptr;
  const int owB = in_w >> 2; // width of the LL plane below level b
  if (!b_last) {
    llp = (ST *)pb.ll[comp] + (size_t)pic * pb.ll_stride[comp] + (size_t)chunk * 2;
    if constexpr (S_::narrow) llp_w = pb.ll_wide[comp] + (size_t)pic * pb.ll_stride[comp] + (size_t)chunk * 2;
  }
  // the lane's place in the LDS images (element offsets): its slice's run, its columns inside a block row
  const int baseA = si * ssA + cl * 4, baseB = si * ssB + cl * 2;
  // Values beyond 16 bits (never with 10-bit DD97; 16-bit samples reach them): the value goes to the wide store at the
  // element's own index, the sentinel into the image.  ONE test per row for all of a lane's values; this is the rare path.
  auto escapes = [&](int *v, auto NBc, auto NVc, int sv, int run0, int bn, int off_r, int clv) __attribute__((always_inline)) {
    constexpr int NB = decltype(NBc)::value, NV = decltype(NVc)::value, H = NV / 2; // bands, values per band and lane
    if constexpr (S_::narrow) {
      // one address for each half of the lane's columns (the same slice's next columns, or the next slice's first), the
      // values at band distance behind it: a compare and a predicated store per value
      int32_t *w0 = wide + mul24z(sv * xs + sp.sx0 + si, slice_coefs) + run0 + off_r + (spl == 1 ? clv : 0);
      int32_t *w1 = spl == 1 ? w0 + H : w0 + slice_coefs;
#pragma unroll
      for (int b = 0; b < NB; ++b, w0 += bn, w1 += bn)
#pragma unroll
        for (int j = 0; j < NV; ++j)
          if (!S_::fits(v[b * NV + j])) { (j < H ? w0 : w1)[j % H] = v[b * NV + j]; v[b * NV + j] = VC2_ST_SENTINEL; }
    }
  };
  // the lane's four values of each of the bands HL, LH, HH of level a: block row r of slice row sv
  auto stageA = [&](ST *img, int r, int sv, const RowT<4> &oe, const RowT<4> &oo) __attribute__((always_inline)) {
    int v[12] = {oe[4], oe[5], oe[6], oe[7], oo[0], oo[1], oo[2], oo[3], oo[4], oo[5], oo[6], oo[7]};
    if constexpr (S_::narrow) {
      int mx = v[0], mn = v[0];
#pragma unroll
      for (int j = 1; j < 12; ++j) { mx = max(mx, v[j]); mn = min(mn, v[j]); }
      if (__builtin_expect(mx > 32767 || mn < -32767, 0)) escapes(v, IC<3>(), IC<4>(), sv, run0A, bnA, r * bswA, cl * 4);
    }
    ST *d = img + baseA + r * bswA;
#pragma unroll
    for (int b = 0; b < 3; ++b, d += bnA) {
      if constexpr (S_::narrow) {
        const unsigned w0 = vc2_pack16(v[4 * b], v[4 * b + 1]), w1 = vc2_pack16(v[4 * b + 2], v[4 * b + 3]);
        if (spl == 1) *(uint2 *)d = make_uint2(w0, w1);
        else { *(unsigned *)d = w0; *(unsigned *)(d + ssA) = w1; } // two slices of two columns each
      } else {
        if (spl == 1) *(int4 *)d = make_int4(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);
        else { *(int2 *)d = make_int2(v[4 * b], v[4 * b + 1]); *(int2 *)(d + ssA) = make_int2(v[4 * b + 2], v[4 * b + 3]); }
      }
    }
  };
  // the lane's two values of each band of level b (LL first when it goes to the store)
  auto stageB = [&](ST *img, int r, int sv, const RowT<2> &oe, const RowT<2> &oo) __attribute__((always_inline)) {
    int v[8] = {oe[0], oe[1], oe[2], oe[3], oo[0], oo[1], oo[2], oo[3]};
    if constexpr (S_::narrow) {
      int mx = v[2], mn = v[2];
#pragma unroll
      for (int j = 3; j < 8; ++j) { mx = max(mx, v[j]); mn = min(mn, v[j]); }
      if (b_last) { mx = max(mx, max(v[0], v[1])); mn = min(mn, min(v[0], v[1])); }
      if (__builtin_expect(mx > 32767 || mn < -32767, 0)) {
        if (b_last) escapes(v, IC<4>(), IC<2>(), sv, run0B, bnB, r * bswB, cl * 2);
        else escapes(v + 2, IC<3>(), IC<2>(), sv, run0B, bnB, r * bswB, cl * 2);
      }
    }
    ST *d = img + baseB + r * bswB;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b == 0 && !b_last) continue;
      if constexpr (S_::narrow) {
        if (spl == 1) *(unsigned *)d = vc2_pack16(v[2 * b], v[2 * b + 1]);
        else { *d = (ST)v[2 * b]; *(d + ssB) = (ST)v[2 * b + 1]; } // two slices of one column each
      } else {
        if (spl == 1) *(int2 *)d = make_int2(v[2 * b], v[2 * b + 1]);
        else { *d = (ST)v[2 * b]; *(d + ssB) = (ST)v[2 * b + 1]; }
      }
      d += bnB;
    }
  };
  // an image (one slice row of the strip's slices) to the slice records, piece by piece.  A trip moves the pieces of
  // `spt` whole slices (a run of up to 64 pieces), or 64 pieces of one slice's longer run: the lane's place in a trip is
  // computed once per flush, a trip only adds strides
  auto flush = [&](const ST *img, int ss, int run0, int run_n, int piece, int sv) __attribute__((always_inline)) {
    const int epp = piece / (int)sizeof(ST), ppr = run_n / epp; // elements per piece, pieces per run
    ST *rec0 = store + mul24z(sv * xs + sp.sx0, slice_coefs) + run0;
    // Runs that are WHOLE 128-byte lines of the first launch leave non-temporally: the slice coder reads them gigabytes later,
    // and 128 UHD pictures take 1.93 - 2.0 instead of 2.29 - 2.31 ms.  Not partial lines: their pieces no longer meet in L2 --
    // the deep levels' launch went from 0.34 to 0.63 ms, the first launch of 32 HD pictures (192-byte runs) from 0.144 to 0.218.
    const bool nt = FIRST && VC2_PAIR_NT && piece == 16 && (((run0 | run_n | slice_coefs) * (int)sizeof(ST)) & 127) == 0;
    if (ppr <= 64) {
      const int spt = 64 / ppr;                                  // slices per trip
      int ls = (int)((float)lane * (1.0f / (float)ppr));         // lane / ppr (exact: both below 2^7, corrected below)
      ls -= ls * ppr > lane ? 1 : 0;
      ls += (ls + 1) * ppr <= lane ? 1 : 0;
      const int e = (lane - ls * ppr) * epp;
      const ST *src = img + ls * ss + e;
      ST *dst = rec0 + mul24z(ls, slice_coefs) + e;
      const int dsrc = spt * ss;
      const size_t ddst = mul24z(spt, slice_coefs);
#pragma unroll 1
      for (int s2 = ls; s2 < sp.nsl; s2 += spt, src += dsrc, dst += ddst) {
        if (ls < spt) {
          if (piece == 16) { if (nt) st_nt(dst, *(const uint4 *)src); else *(uint4 *)dst = *(const uint4 *)src; }
          else if (piece == 8) *(uint2 *)dst = *(const uint2 *)src;
          else *(unsigned *)dst = *(const unsigned *)src;
        }
      }
    } else {
#pragma unroll 1
      for (int s2 = 0; s2 < sp.nsl; ++s2) {
        const ST *src = img + s2 * ss;
        ST *dst = rec0 + mul24z(s2, slice_coefs);
#pragma unroll 1
        for (int q = lane; q < ppr; q += 64) {
          if (piece == 16) { if (nt) st_nt(dst + q * epp, *(const uint4 *)(src + q * epp)); else *(uint4 *)(dst + q * epp) = *(const uint4 *)(src + q * epp); }
          else if (piece == 8) *(uint2 *)(dst + q * epp) = *(const uint2 *)(src + q * epp);
          else *(unsigned *)(dst + q * epp) = *(const unsigned *)(src + q * epp);
        }
      }
    }
  };

  // ---- the walk.  Level a's pair m enters at phase U0 = (m - m0) mod 8; its completed pair is k = m - OFFL; the LL rows
  // of k = 2 m1 and 2 m1 + 1 are the even / odd row of level b's pair m1 (the walk starts at an even m0, so the parity of
  // k is a compile-time constant of the phase).
  EA engA;
  EB engB;
  engA.clear();
  engB.clear();
  const int kA = sp.kA, kB = sp.kB, kA1 = kA >> 1, kB1 = kB >> 1;
  const int want = (kA + 3 * SD) & ~1;        // first pair of level a the segment's results depend on
  const bool top = want <= 0;
  const bool bottom = kB >= npA;              // the walk ends with the plane's last pair and the steps below it
  // a bottom walk starts a whole number of blocks above the plane's end, so that the steps below the plane always run
  // at phase 0 (the host leaves room: vc2_pair_applicable); every other walk where its results begin
  const int m0 = top ? 0 : bottom ? npA - ((npA - want + 7) & ~7) : want;
  const int mend = bottom ? npA : kB + 3 * OFFL; // level b's last pair needs 2 * OFFL more LL rows, those OFFL more pairs
  const int mload = npA - 1;
#pragma unroll
  for (int u = 0; u < PFP; ++u) fetch(min(m0 + u, mload), u);
  static_assert(8 % PFP == 0, "the prefetch ring shares the unrolled walk");
  // What the walk has staged and not yet stored (all wave-uniform): slice rows [stA, dnA) of level a, [stB, dnB) of level b,
  // LL rows [stL, dnL) of level b.  A burst is due when level a completes the last slice row of a group (or of the segment);
  // it happens at the top of the next iteration, where the rows just consumed have arrived and the next prefetch goes out
  // right behind it: every burst is one drain of the wavefront's memory operations (loads and stores share one in-order
  // counter), so everything that is ready leaves with it -- level b's images and LL rows wait for level a's burst.
  const int sv0 = kA >> lbshA, sv1 = kB >> lbshA;
  int stA = sv0, dnA = sv0, stB = sv0, dnB = sv0, stL = kA1, dnL = kA1;
  int slotBs = 0, slotBf = 0;                    // level b's ring: the slot being staged, the first slot not yet stored
  bool due = false;
  int llE[4] = {0, 0, 0, 0};                     // the LL row of the last even k
  int mb = m0;

  auto burst = [&]() __attribute__((always_inline)) {
    wave_sync();
#pragma unroll 1
    for (; stA < dnA; ++stA) if (!(dskip & 256)) flush(imgA + ((stA - sv0) & (grpA - 1)) * szA, ssA, run0A, 3 * bnA, pieceA, stA);
#pragma unroll 1
    for (; stB < dnB; ++stB) {
      if (!(dskip & 64)) flush(imgB + slotBf * szB, ssB, run0B, (b_last ? 4 : 3) * bnB, pieceB, stB);
      slotBf = slotBf == grpA ? 0 : slotBf + 1;
    }
    if (!b_last) {
#pragma unroll 1
      for (; stL < dnL; ++stL) {
        if (own && !(dskip & 128)) {
          ST *d = llp + mul24z(stL, owB);
          if constexpr (S_::narrow) *(unsigned *)d = ring_ll[(stL & (nll - 1)) * 64 + lane];
          else *(uint2 *)d = *(const uint2 *)(ring_ll + ((stL & (nll - 1)) * 64 + lane) * 2);
        }
      }
    }
    wave_sync();
    due = false;
  };
  auto deferred = [&]() __attribute__((always_inline)) { if (__builtin_expect(due, 0)) burst(); }; // (cold: placed out of the walk's straight line)
  // the pair k1 of level b that engB has just completed at phase UB
  auto emitB = [&](auto UBc, int k1) __attribute__((always_inline)) {
    constexpr int UB = decltype(UBc)::value;
    if (k1 >= kA1 && k1 < kB1) {
      const RowT<2> &oe = engB.template out<UB>(false), &oo = engB.template out<UB>(true);
      const int r = k1 & (bshB - 1), sv = k1 >> lbshB;
      if (!(dskip & 16)) {
        if (!b_last) { // the LL row into its ring (beyond 16 bits: the wide plane at once, the sentinel into the ring)
          int a0 = oe[0], a1 = oe[1];
          if constexpr (S_::narrow) {
            if (__builtin_expect(own && (!S_::fits(a0) || !S_::fits(a1)), 0)) {
              int32_t *w = llp_w + mul24z(k1, owB);
              if (!S_::fits(a0)) { w[0] = a0; a0 = VC2_ST_SENTINEL; }
              if (!S_::fits(a1)) { w[1] = a1; a1 = VC2_ST_SENTINEL; }
            }
#if !(defined(VC2_PAIR_X) && (VC2_PAIR_X & 2)) // (timing experiments: compile-time removal of one part of level b's output path)
            ring_ll[(k1 & (nll - 1)) * 64 + lane] = vc2_pack16(a0, a1);
#endif
          } else *(uint2 *)(ring_ll + ((k1 & (nll - 1)) * 64 + lane) * 2) = make_uint2((unsigned)a0, (unsigned)a1);
          dnL = k1 + 1;
        }
#if !(defined(VC2_PAIR_X) && (VC2_PAIR_X & 1))
        if (own) stageB(imgB + slotBs * szB, r, sv, oe, oo);
#endif
#if !(defined(VC2_PAIR_X) && (VC2_PAIR_X & 4))
        if (r == bshB - 1) {
          dnB = sv + 1;
          slotBs = slotBs == grpA ? 0 : slotBs + 1;
          if (sv + 1 == sv1) due = true; // (level b's last slice row of the segment: nothing of level a is left to wait for)
        }
#endif
      }
    }
  };
  // one pair of level a.  BM: 0 steady, 1 / 2 the first / second block of a walk that starts at the plane's top (level a
  // fills its rings in iterations 0..3, level b with its pairs 0..3); DRAIN: below the plane's last pair (D = U0 < OFFL)
  auto iter = [&](auto U0c, auto BMc, auto DRc) __attribute__((always_inline)) {
    constexpr int U0 = decltype(U0c)::value, BM = decltype(BMc)::value;
    constexpr bool DRAIN = decltype(DRc)::value;
    if constexpr (!DRAIN || U0 < OFFL) {
      constexpr int UA = U0 & 3;
      constexpr int MODEA = DRAIN ? 2 : (BM == 1 && U0 < 4) ? 1 : 0;
      const int m = DRAIN ? npA + U0 : mb + U0;
      RowT<4> re, ro;
      if constexpr (!DRAIN) {
        convert(min(m, mload), U0 % PFP, 0, re);
        convert(min(m, mload), U0 % PFP, 1, ro);
      }
      deferred();
      if constexpr (!DRAIN) {
        fetch(min(m + PFP, mload), U0 % PFP);
        h_lift<K, false, 4>(re, redge);
        h_lift<K, false, 4>(ro, redge);
      }
      engA.template step<UA, MODEA, (DRAIN ? U0 : UA)>(m, npA, re, ro);
      const int k = m - OFFL;
      const RowT<4> &oe = engA.template out<UA>(false), &oo = engA.template out<UA>(true);
      if (k >= kA && k < kB) {
        const int r = k & (bshA - 1), sv = k >> lbshA;
        if (!(dskip & 32)) {
          if (own) stageA(imgA + ((sv - sv0) & (grpA - 1)) * szA, r, sv, oe, oo);
          if (r == bshA - 1) {
            dnA = sv + 1;
            if (((sv + 1 - sv0) & (grpA - 1)) == 0 || sv + 1 == sv1) due = true;
          }
        }
      }
      constexpr bool KODD = ((U0 - OFFL) & 1) != 0;
      if constexpr (!KODD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) llE[j] = oe[j];
      } else {
        constexpr int J = fdiv2(U0 - OFFL - 1);      // level b's pair, counted from the block's (the plane end's) first
        constexpr int UB = ((J % 4) + 4) % 4;
        constexpr bool SKIP = BM == 1 && J < 0;      // pairs above the plane
        constexpr int MODEB = (BM == 1 && J >= 0 && J < 4) || (BM == 2 && J < 0) ? 1 : 0;
        if (!SKIP && !(dskip & 8)) {
          const int m1 = ((DRAIN ? npA : mb) >> 1) + J;
          RowT<2> re1, ro1; // (even columns first)
          re1[0] = (int)((unsigned)llE[0] << ACC); re1[1] = (int)((unsigned)llE[2] << ACC);
          re1[2] = (int)((unsigned)llE[1] << ACC); re1[3] = (int)((unsigned)llE[3] << ACC);
          ro1[0] = (int)((unsigned)oe[0] << ACC); ro1[1] = (int)((unsigned)oe[2] << ACC);
          ro1[2] = (int)((unsigned)oe[1] << ACC); ro1[3] = (int)((unsigned)oe[3] << ACC);
          h_lift<K, false, 2>(re1, redge);
          h_lift<K, false, 2>(ro1, redge);
          engB.template step<UB, MODEB>(m1, npB, re1, ro1);
          emitB(IC<UB>(), m1 - OFFL);
        }
      }
    }
  };
  auto block = [&](auto BMc, auto DRc) __attribute__((always_inline)) {
    iter(IC<0>(), BMc, DRc); iter(IC<1>(), BMc, DRc); iter(IC<2>(), BMc, DRc); iter(IC<3>(), BMc, DRc);
    iter(IC<4>(), BMc, DRc); iter(IC<5>(), BMc, DRc); iter(IC<6>(), BMc, DRc); iter(IC<7>(), BMc, DRc);
  };
  if (top) { // (the host admits planes of at least 24 row pairs: both blocks lie inside the plane)
    block(IC<1>(), BC<false>());
    mb += 8;
    block(IC<2>(), BC<false>());
    mb += 8;
  }
  const int prio0 = (int)(blockIdx.x >> 10);
  for (; mb < mend; mb += 8) {
    if (st_prio) pair_prio(st_prio, prio0, mb >> 3);
    block(IC<0>(), BC<false>());
  }
  if (bottom) {
    block(IC<0>(), BC<true>()); // level a below the plane: OFFL steps, level b's last pairs among them
    // level b below ITS last pair
    auto drainB = [&](auto Dc) __attribute__((always_inline)) {
      constexpr int D = decltype(Dc)::value;
      if constexpr (D < OFFL) {
        deferred();
        RowT<2> none;
#pragma unroll
        for (int j = 0; j < 4; ++j) none[j] = 0;
        engB.template step<D, 2, D>(npB + D, npB, none, none);
        emitB(IC<D>(), npB + D - OFFL);
      }
    };
    drainB(IC<0>()); drainB(IC<1>()); drainB(IC<2>()); drainB(IC<3>());
  }
  burst(); // whatever is left (level b's last rows complete behind level a's last burst)
#undef p
#undef pb
}

// ------------------------------------------------------------------------------------------
// inverse: levels b = a + 1 (the coarser one, first) and a
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int dequant_fullp(int v, int qf, int off) { // scale(), Quantisation.cpp:86-95, literally
  if (v == 0) return 0;
  const unsigned mag = v < 0 ? 0u - (unsigned)v : (unsigned)v;
  int a = (int)(mag * (unsigned)qf);
  if (a > 0) a = (int)((unsigned)a + (unsigned)off);
  a = (int)((unsigned)a + 2u);
  a /= 4;
  return v < 0 ? (int)(0u - (unsigned)a) : a;
}

// Level b's pair m1 = m / 2 + OFFL enters its engine in the iteration of level a's even pair m: the pair it completes,
// k1 = m / 2, is -- after the horizontal inverse lifting and the rounding -- the LL rows of level a's pairs m and m + 1.
// Above the walk's first pair level b runs a prologue of its own (its filter run-in); in the last block of a walk that
// ends with the plane, level b is already below ITS last pair.
#ifndef VC2_PAIR_WPE_INV
#define VC2_PAIR_WPE_INV 2
#endif
#ifndef VC2_PAIR_PFI
#define VC2_PAIR_PFI 2
#endif
constexpr int PFQ = VC2_PAIR_PFI; // row pairs of level a's bands prefetched ahead (inverse)
// SPL2: some component's slices are 4 samples wide at level a (a lane spans two slices): its own instantiation, so that
// the common one (every BASELINE configuration's two finest levels) carries none of that code
template <int K, bool FINAL, class ST, bool SPL2>
__global__ __launch_bounds__(64, (K == VC2HIP_DD137 || SPL2 ? 2 : VC2_PAIR_WPE_INV)) void k_inv_pair(const PairParams pp) {
  using S_ = St<ST>;
  using EA = VEng<K, true, 4>;
  using EB = VEng<K, true, 2>;
  using T = VT<K, true>;
  constexpr int OFFL = T::OFFL, SD = T::sum_dmin();
  constexpr int ACC = WT<K>::accuracy;
  __shared__ int qtab[360]; // quant_factor / quant_offset / fast-path limit by adjusted index (as k_inv_stream)
  __shared__ int rare[12 * 64]; // the rare paths' values, per lane (escapes of the 16-bit store; values beyond the dequantiser's fast domain)
  // Level b's engine is used in every second iteration; between its steps its rows wait in LDS (VEng::park): its ~36
  // registers are what separates this kernel from three wavefronts per SIMD
  constexpr bool PARK = !SPL2 && VC2_PAIR_WPE_INV >= 3;
  __shared__ __attribute__((aligned(16))) int parked[PARK ? VEng<K, true, 2>::park_rows() * 64 * 4 : 4];
#define p pp.a
#define pb pp.b
  const int lane = threadIdx.x & 63;
  int comp, pic;
  Strip sp;
  if (!strip_of_block_h<pair_halo<K>()>(p, comp, pic, sp)) return;
  for (int i = lane; i < 120; i += 64) {
    const int qf = c_qsp.qf[i], off = c_qsp.off[i];
    qtab[i] = qf; qtab[120 + i] = off;
    qtab[240 + i] = (qf > 0 && qf < (1 << 24)) ? (int)min(((1u << 25) - (unsigned)off - 8u) / (unsigned)qf, 0x7FFFFFu) : -1;
  }
  wave_sync();
  // (every member of the argument read once, here: see k_fwd_pair)
  const int spl = SPL2 ? pp.spl[comp] : 1;
  const int xs = p.xs, st_prio = p.st_prio, dequant = p.dequant;
#ifdef VC2HIP_ABLATE // timing experiments (VC2HIP_DEBUG_SKIP): 8 level b not at all, 16 its values not unpacked / dequantised, 64 not loaded, 128 no vertical step, 256 no horizontal lifting
  const int dskip = p.debug_skip;
#else
  constexpr int dskip = 0;
#endif
  const int out_h = p.in_h[comp], out_w = p.in_w[comp], npA = out_h >> 1, npB = npA >> 1, owA = out_w >> 1, owB = out_w >> 2;
  const int chunk = min(sp.c0 + lane, (out_w >> 3) - 1);
  const bool own = lane >= sp.lo && lane < sp.hi;
  const bool redge = sp.c0 + lane == (out_w >> 3) - 1 && lane != 63;
  const int bshA = p.fh[comp] >> 1, bswA = p.fw[comp] >> 1, lbshA = ilog2d(bshA), bnA = bshA * bswA;
  const int bshB = bshA >> 1, bswB = bswA >> 1, lbshB = lbshA - 1, bnB = bshB * bswB;
  const int llps = p.st_llps[comp];
  const int sx = spl == 1 ? chunk >> llps : chunk * 2;                 // the lane's (first) slice
  const int cl = spl == 1 ? chunk & ((1 << llps) - 1) : 0;             // its chunk inside the slice
  const int b_last = pb.ll_from_store;                                 // level b's LL comes from store band 0 (dequantised)
  const int rsA = p.rec_stride[comp], rsB = pb.rec_stride[comp];       // slice records, or the record heads (HeadSplit)
  const int run0A = p.coef_off[comp] + bnA + cl * 4, run0B = pb.coef_off[comp] + (b_last ? 0 : bnB) + cl * 2;
  const long long bpA = p.bp_base[comp], bpB = pb.bp_base[comp];       // the level's band planes (-1: in the records)
  const ST *store = (const ST *)p.store + (size_t)pic * p.store_stride;
  const int32_t *wide = S_::narrow ? p.store_wide + (size_t)pic * p.store_stride : nullptr;
  const int32_t *qidx = p.qidx ? p.qidx + (size_t)pic * p.ys * xs : nullptr;
  // Addresses: a wave-uniform base (the picture's store, plane, output) plus a 32-bit byte offset per lane -- the memory
  // instructions then take the base from scalar registers and one register of offset instead of a 64-bit pointer per lane
  // (the host admits this kernel only where a picture's store and planes stay below 4 GiB)
  const ST *llp = nullptr;     // level b's LL plane (this picture)
  const int32_t *llp_w = nullptr;
  if (!b_last) {
    llp = (const ST *)pb.ll[comp] + (size_t)pic * pb.ll_stride[comp];
    if constexpr (S_::narrow) llp_w = pb.ll_wide[comp] + (size_t)pic * pb.ll_stride[comp];
  }
  auto ldq = [&](const void *base, unsigned elem, auto tag) __attribute__((always_inline)) {
    using Q = decltype(tag);
    return *(const Q *)((const char *)base + (size_t)(elem * (unsigned)sizeof(ST)));
  };
  const int qmA = p.band, qmB = pb.band; // band index of HL at the level (quantisation matrix)
  const int qm0 = p.qmatrix[0];
  int qmxA[3], qmxB[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) { qmxA[b] = p.qmatrix[qmA + b]; qmxB[b] = p.qmatrix[qmB + b]; }

  // ---- input.  Element index (from the picture's store) of the lane's coefficients of band b (1..3; 0 = LL at the last
  // level) in band row m: in the level's band planes simply row m, the lane's columns; in the records / heads the block
  // row of the lane's slice (the second half of the lane's columns is the NEXT slice's when spl == 2)
  const unsigned bpA32 = (unsigned)bpA + (unsigned)chunk * 4, bpB32 = (unsigned)bpB + (unsigned)chunk * 2;
  auto atA = [&](int m, int b) __attribute__((always_inline)) -> unsigned {
    if (bpA >= 0) return bpA32 + __umul24((unsigned)((b - 1) * npA + m), (unsigned)owA);
    const int sv = m >> lbshA, r = m & (bshA - 1);
    return __umul24((unsigned)(sv * xs + sx), (unsigned)rsA) + (unsigned)(run0A + (b - 1) * bnA + r * bswA);
  };
  auto atB = [&](int m1, int b) __attribute__((always_inline)) -> unsigned {
    if (bpB >= 0 && b > 0) return bpB32 + __umul24((unsigned)((b - 1) * npB + m1), (unsigned)owB);
    const int sv = m1 >> lbshB, r = m1 & (bshB - 1);
    return __umul24((unsigned)(sv * xs + sx), (unsigned)rsB) + (unsigned)(run0B + (b - (b_last ? 0 : 1)) * bnB + r * bswB);
  };
  typedef typename std::conditional<S_::narrow, uint2, uint4>::type Q4;    // four store elements
  typedef typename std::conditional<S_::narrow, unsigned, uint2>::type Q2; // two
  typedef typename std::conditional<S_::narrow, unsigned short, unsigned>::type Q1;
  Q4 bqA[PFQ][3];
  Q2 bqB[4]; // [0] = LL
  const bool splitA = spl == 2 && bpA < 0, splitB = spl == 2 && bpB < 0; // the lane's columns lie in two slices' records
  auto fetchA = [&](int m, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 1; b < 4; ++b) {
      const unsigned q = atA(m, b);
      if (!splitA) bqA[slot][b - 1] = ldq(store, q, Q4());
      else {
        const Q2 lo = ldq(store, q, Q2()), hi = ldq(store, q + (unsigned)rsA, Q2());
        if constexpr (S_::narrow) bqA[slot][b - 1] = make_uint2(lo, hi);
        else bqA[slot][b - 1] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
  };
  auto fetchB = [&](int m1) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b == 0 && !b_last) { bqB[0] = ldq(llp, __umul24((unsigned)m1, (unsigned)owB) + (unsigned)chunk * 2, Q2()); continue; }
      const unsigned q = atB(m1, b);
      const bool split = b == 0 ? spl == 2 : splitB;
      if (!split) bqB[b] = ldq(store, q, Q2());
      else {
        const Q1 lo = ldq(store, q, Q1()), hi = ldq(store, q + (unsigned)rsB, Q1());
        if constexpr (S_::narrow) bqB[b] = (unsigned)lo | ((unsigned)hi << 16);
        else bqB[b] = make_uint2(lo, hi);
      }
    }
  };
  // quantiser constants of the lane's slice(s) in a slice row: level a keeps factor / offset / fast-path limit of its three
  // bands in registers (one slice per lane); everything else -- level b, and lanes that span two slices -- looks them up
  // in the LDS table by adjusted index at each use
  constexpr int NS = SPL2 ? 2 : 1; // sets of constants: the lane's first / second slice (equal where the lane lies in one slice)
  // Where they live: in the common instantiation in LDS, a row of 28 dwords per lane (level a: factors, offsets, limits of
  // its three bands as three 16-byte reads per iteration; level b: of LL + three bands, read per pair) -- the 21 registers
  // they would take are what separates this kernel from three wavefronts per SIMD, and its time is its instruction stream
  // at the rate its wavefronts issue.  (28: the lanes' rows then start in different banks.)
  constexpr bool QL = false; // (measured: the constants' registers are not where the pressure peaks -- 190 registers either way)
  constexpr int QROW = 28;
  __shared__ __attribute__((aligned(16))) int qrow[QL ? QROW * 64 : 4];
  int *const qmine = qrow + (QL ? lane * QROW : 0);
  int qfA[NS][3], qoA[NS][3], qlA[NS][3];
  int qfB[NS][4], qoB[NS][4], qlB[NS][4]; // level b: [0] = LL (the last level), then HL, LH, HH
  int qsA[2] = {0, 0}, qsB[2] = {0, 0}; // quantiser index of the lane's first / second slice in level a's / b's current slice row
  int svA_have = -1, svB_have = -1;
  auto check_q = [&](int q, int qm) __attribute__((always_inline)) { if (dequant && q - qm > 119) atomicOr(p.err, VC2_DEVERR_QINDEX); };
  auto load_qA = [&](int sv) __attribute__((always_inline)) {
    qsA[0] = dequant ? qidx[sv * xs + sx] : 0;
    qsA[1] = dequant && spl == 2 ? qidx[sv * xs + sx + 1] : qsA[0];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      check_q(max(qsA[0], qsA[1]), qmxA[b]);
#pragma unroll
      for (int h = 0; h < NS; ++h) {
        const int aq = min(max(qsA[h] - qmxA[b], 0), 119);
        if constexpr (QL) { qmine[b] = qtab[aq]; qmine[4 + b] = qtab[120 + aq]; qmine[8 + b] = qtab[240 + aq]; }
        else { qfA[h][b] = qtab[aq]; qoA[h][b] = qtab[120 + aq]; qlA[h][b] = qtab[240 + aq]; }
      }
    }
  };
  auto fetch_qA = [&]() __attribute__((always_inline)) { // (QL) this iteration's copy of level a's constants
    if constexpr (QL) {
      const int4 f = *(const int4 *)qmine, o = *(const int4 *)(qmine + 4), l = *(const int4 *)(qmine + 8);
      qfA[0][0] = f.x; qfA[0][1] = f.y; qfA[0][2] = f.z; qoA[0][0] = o.x; qoA[0][1] = o.y; qoA[0][2] = o.z;
      qlA[0][0] = l.x; qlA[0][1] = l.y; qlA[0][2] = l.z;
    }
  };
  auto fetch_qB = [&]() __attribute__((always_inline)) {
    if constexpr (QL) {
      const int4 f = *(const int4 *)(qmine + 12), o = *(const int4 *)(qmine + 16), l = *(const int4 *)(qmine + 20);
      qfB[0][0] = f.x; qfB[0][1] = f.y; qfB[0][2] = f.z; qfB[0][3] = f.w; qoB[0][0] = o.x; qoB[0][1] = o.y; qoB[0][2] = o.z; qoB[0][3] = o.w;
      qlB[0][0] = l.x; qlB[0][1] = l.y; qlB[0][2] = l.z; qlB[0][3] = l.w;
    }
  };
  auto load_qB = [&](int sv) __attribute__((always_inline)) {
    qsB[0] = dequant ? qidx[sv * xs + sx] : 0;
    qsB[1] = dequant && spl == 2 ? qidx[sv * xs + sx + 1] : qsB[0];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int qm = b == 0 ? qm0 : qmxB[b - 1];
      if (b > 0 || b_last) check_q(max(qsB[0], qsB[1]), qm);
#pragma unroll
      for (int h = 0; h < NS; ++h) {
        const int aq = min(max(qsB[h] - qm, 0), 119);
        if constexpr (QL) { qmine[12 + b] = qtab[aq]; qmine[16 + b] = qtab[120 + aq]; qmine[20 + b] = qtab[240 + aq]; }
        else { qfB[h][b] = qtab[aq]; qoB[h][b] = qtab[120 + aq]; qlB[h][b] = qtab[240 + aq]; }
      }
    }
  };
  // The lane's values of one load.  Hot path: unpack, ONE test for escapes of the 16-bit store (the sentinel is the smallest
  // 16-bit value) and for the dequantiser's fast domain, five instructions per value.  Everything else -- an escape (its value
  // is in the wide array at the element's own index), a value beyond the fast domain, lanes that span two slices -- takes the
  // rare path: the values go through an LDS row per lane and a ROLLED loop with scale() literally (the unrolled form of these
  // paths was two thirds of the kernel's code, and the walk's unrolled block no longer fitted the instruction cache).
  auto rare_path = [&](int *v, auto NVc, auto atf, int m, const int *qs, int band0) __attribute__((always_inline)) {
    constexpr int NV = decltype(NVc)::value, N = 3 * NV + (NV == 2 ? 2 : 0); // values per band; values in all
#pragma unroll
    for (int k = 0; k < N; ++k) rare[k * 64 + lane] = v[k];
    wave_sync();
#pragma unroll 1
    for (int k = 0; k < N; ++k) {
      const int b = NV == 2 ? k / 2 : k / 4 + 1, j = k % NV; // band (0 = level b's LL), value inside the lane's load
      int x = rare[k * 64 + lane];
      if (NV == 2 && b == 0 && !b_last) { // level b's LL from its plane
        if constexpr (S_::narrow) if (x == VC2_ST_SENTINEL) x = llp_w[mul24z(m, owB) + (size_t)chunk * 2 + j];
      } else {
        const bool split = NV == 2 ? (b == 0 ? spl == 2 : splitB) : splitA; // the lane's second half in the next slice's record
        if constexpr (S_::narrow) if (x == VC2_ST_SENTINEL) {
          const int32_t *wq = wide + atf(m, b);
          x = split && j >= NV / 2 ? wq[(NV == 2 ? rsB : rsA) + j - NV / 2] : wq[j];
        }
        if (dequant) {
          const int qm = b == 0 ? qm0 : p.qmatrix[band0 + b - 1];
          const int aq = min(max(qs[spl == 2 ? j / (NV / 2) : 0] - qm, 0), 119);
          x = dequant_fullp(x, qtab[aq], qtab[120 + aq]);
        }
      }
      rare[k * 64 + lane] = x;
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = rare[k * 64 + lane];
  };
  auto unpackA = [&](int m, int slot, int (&v)[12]) __attribute__((always_inline)) {
    bool hot = true;
    if (dequant) fetch_qA();
    if constexpr (S_::narrow) {
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const uint2 w = bqA[slot][b];
        v[4 * b] = vc2_lo16(w.x); v[4 * b + 1] = vc2_hi16(w.x); v[4 * b + 2] = vc2_lo16(w.y); v[4 * b + 3] = vc2_hi16(w.y);
      }
      int lowest = v[0];
#pragma unroll
      for (int k = 1; k < 12; ++k) lowest = min(lowest, v[k]);
      hot = lowest != VC2_ST_SENTINEL;
    } else {
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const uint4 w = bqA[slot][b];
        v[4 * b] = (int)w.x; v[4 * b + 1] = (int)w.y; v[4 * b + 2] = (int)w.z; v[4 * b + 3] = (int)w.w;
      }
    }
    if (dequant) {
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) { // (the lane's first / second pair of columns: one slice's limit each when it spans two)
          const int mx = max(v[4 * b + 2 * h], v[4 * b + 2 * h + 1]), mn = min(v[4 * b + 2 * h], v[4 * b + 2 * h + 1]);
          hot &= mx <= qlA[h % NS][b] && mn >= -qlA[h % NS][b];
        }
    }
    if (__builtin_expect(hot, 1)) {
      if (dequant) { // scale(), Quantisation.cpp:86-95, inside its domain: sign(v) * ((|v| * factor + offset + 2) >> 2)
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          const int x = v[k], sg = min(max(x, -1), 1);
          const unsigned t = (__umul24((unsigned)__mul24(x, sg), (unsigned)qfA[((k >> 1) & 1) % NS][k >> 2]) + (unsigned)(qoA[((k >> 1) & 1) % NS][k >> 2] + 2)) >> 2;
          v[k] = __mul24((int)t, sg);
        }
      }
    } else rare_path(v, IC<4>(), atA, m, qsA, qmA);
  };
  auto unpackB = [&](int m1, int (&v)[8]) __attribute__((always_inline)) { // [0..1] LL, then HL, LH, HH
    bool hot = true;
    if (dequant) fetch_qB();
    if constexpr (S_::narrow) {
#pragma unroll
      for (int b = 0; b < 4; ++b) { v[2 * b] = vc2_lo16(bqB[b]); v[2 * b + 1] = vc2_hi16(bqB[b]); }
      int lowest = v[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) lowest = min(lowest, v[k]);
      hot = lowest != VC2_ST_SENTINEL;
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) { v[2 * b] = (int)bqB[b].x; v[2 * b + 1] = (int)bqB[b].y; }
    }
    if (dequant) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (b == 0 && !b_last) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) hot &= v[2 * b + h] <= qlB[h % NS][b] && v[2 * b + h] >= -qlB[h % NS][b];
      }
    }
    if (__builtin_expect(hot, 1)) {
      if (dequant) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (k < 2 && !b_last) continue;
          const int x = v[k], sg = min(max(x, -1), 1);
          const unsigned t = (__umul24((unsigned)__mul24(x, sg), (unsigned)qfB[(k & 1) % NS][k >> 1]) + (unsigned)(qoB[(k & 1) % NS][k >> 1] + 2)) >> 2;
          v[k] = __mul24((int)t, sg);
        }
      }
    } else rare_path(v, IC<2>(), atB, m1, qsB, qmB);
  };

  // ---- output rows of level a (as k_inv_stream)
  const int lim_h = FINAL ? p.pic_h[comp] : out_h;
  const int clip_lo = p.clip_lo, clip_hi = p.clip_hi, sample_offset = p.sample_offset, sample_shift = p.sample_shift;
  uint8_t *rawo = nullptr; // (this picture's output plane; the lane's columns are part of the offset)
  ST *lvl = nullptr;
  int32_t *lvl_w = nullptr;
  if constexpr (FINAL) rawo = (uint8_t *)p.plane[comp] + (size_t)pic * p.plane_stride[comp];
  else {
    lvl = (ST *)p.plane[comp] + (size_t)pic * p.plane_stride[comp];
    if constexpr (S_::narrow) lvl_w = p.plane_wide[comp] + (size_t)pic * p.plane_stride[comp];
  }
  const unsigned ocol = (unsigned)chunk * 8; // first output column of the lane
  constexpr int OW = (FINAL || S_::narrow) ? 4 : 8;
  struct Pend { unsigned w[OW]; };
  auto make_out = [&](int y, RowT<4> &r, Pend &o) __attribute__((always_inline)) {
    h_lift<K, true, 4>(r, redge);
    int s[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[2 * k] = r[k]; s[2 * k + 1] = r[4 + k]; }
    if (ACC) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] = (s[k] + (1 << (ACC > 0 ? ACC - 1 : 0))) >> ACC;
    }
    if constexpr (FINAL) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned a = (unsigned)(min(max(s[2 * k], clip_lo), clip_hi) + sample_offset) << sample_shift;
        const unsigned b = (unsigned)(min(max(s[2 * k + 1], clip_lo), clip_hi) + sample_offset) << sample_shift;
        o.w[k] = __builtin_amdgcn_perm(b, a, 0x04050001u); // the low halves of a, b as big-endian 16-bit words (pack and swap in one v_perm)
      }
    } else if constexpr (S_::narrow) {
      const int mx = max(max(max(s[0], s[1]), max(s[2], s[3])), max(max(s[4], s[5]), max(s[6], s[7])));
      const int mn = min(min(min(s[0], s[1]), min(s[2], s[3])), min(min(s[4], s[5]), min(s[6], s[7])));
      if (__builtin_expect((mx > 32767 || mn < -32767) && own && y < lim_h, 0)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) if (!S_::fits(s[k])) { lvl_w[mul24z(y, out_w) + ocol + k] = s[k]; s[k] = VC2_ST_SENTINEL; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) o.w[k] = vc2_pack16(s[2 * k], s[2 * k + 1]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o.w[k] = (unsigned)s[k];
    }
  };
  auto put_out = [&](int y, const Pend &o) __attribute__((always_inline)) {
    if (!own || y >= lim_h) return;
    const unsigned e = __umul24((unsigned)y, (unsigned)out_w) + ocol; // element (sample) index inside the plane
    if constexpr (FINAL) *(uint4 *)(rawo + (size_t)(e * 2u)) = make_uint4(o.w[0], o.w[1], o.w[2], o.w[3]);
    else if constexpr (S_::narrow) *(uint4 *)((char *)lvl + (size_t)(e * 2u)) = make_uint4(o.w[0], o.w[1], o.w[2], o.w[3]);
    else {
      *(uint4 *)((char *)lvl + (size_t)(e * 4u)) = make_uint4(o.w[0], o.w[1], o.w[2], o.w[3]);
      *(uint4 *)((char *)lvl + (size_t)(e * 4u + 16u)) = make_uint4(o.w[OW - 4], o.w[OW - 3], o.w[OW - 2], o.w[OW - 1]);
    }
  };

  // ---- the walk
  EA engA;
  EB engB;
  engA.clear();
  engB.clear();
  if constexpr (PARK) {
#pragma unroll
    for (int r = 0; r < EB::park_rows(); ++r) *(int4 *)(parked + (r * 64 + lane) * 4) = make_int4(0, 0, 0, 0);
  }
  const int kA = sp.kA, kB = sp.kB;
  const int wantA = kA + SD;                       // first pair of level a the segment's rows depend on ...
  const int wantB = (wantA >> 1) + SD;             // ... and of level b (its completed pair wantA / 2 gives level a's LL rows)
  const bool top = wantB <= 0;
  const bool bottom = kB >= npA;
  const int m0e = wantA & ~1;
  const int m0 = top ? 0 : bottom ? npA - ((npA - m0e + 7) & ~7) : m0e;
  const int mend = bottom ? npA : kB + OFFL;
  const int mloadA = npA - 1, mloadB = npB - 1;
  int pend_k = -1;
  Pend pe, po;
#pragma unroll
  for (int k = 0; k < OW; ++k) { pe.w[k] = 0; po.w[k] = 0; }
  int llO[4] = {0, 0, 0, 0}; // level a's LL row of the odd pair that follows
  int mb = m0;

  // one pair of level b: m1 enters at phase UB (MODE as VEng); the pair it completes leaves as level a's two LL rows
  auto stepB = [&](auto UBc, auto MODEc, auto Dc, int m1, bool prefetch, int (&llE)[4]) __attribute__((always_inline)) {
    constexpr int UB = decltype(UBc)::value, MODE = decltype(MODEc)::value, D = decltype(Dc)::value;
    RowT<2> re1, ro1;
    if constexpr (MODE != 2) {
      const int ml = min(max(m1, 0), mloadB), sv = ml >> lbshB;
      if (__builtin_expect(sv != svB_have, 0)) { load_qB(sv); svB_have = sv; }
      int v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (!(dskip & 16)) unpackB(ml, v);
      re1[0] = v[0]; re1[1] = v[1]; re1[2] = v[2]; re1[3] = v[3];
      ro1[0] = v[4]; ro1[1] = v[5]; ro1[2] = v[6]; ro1[3] = v[7];
      if (prefetch && !(dskip & 64)) fetchB(min(max(m1 + 1, 0), mloadB));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { re1[j] = 0; ro1[j] = 0; }
    }
    if constexpr (PARK) engB.template unpark<((UB + 3) & 3)>(parked, lane);
    if (!(dskip & 128)) engB.template step<UB, MODE, D>(m1, npB, re1, ro1);
    RowT<2> oe = engB.template out<UB>(false), oo = engB.template out<UB>(true);
    if constexpr (PARK) engB.template park<UB>(parked, lane);
    if (!(dskip & 256)) {
      h_lift<K, true, 2>(oe, redge);
      h_lift<K, true, 2>(oo, redge);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int a = oe[j], b = oe[2 + j], c = oo[j], d = oo[2 + j];
      if (ACC) { a = (a + (1 << (ACC > 0 ? ACC - 1 : 0))) >> ACC; b = (b + (1 << (ACC > 0 ? ACC - 1 : 0))) >> ACC; c = (c + (1 << (ACC > 0 ? ACC - 1 : 0))) >> ACC; d = (d + (1 << (ACC > 0 ? ACC - 1 : 0))) >> ACC; }
      llE[2 * j] = a; llE[2 * j + 1] = b; llO[2 * j] = c; llO[2 * j + 1] = d;
    }
  };
  // BM: 0 steady, 1 the first block of a walk that starts at the plane's top, 3 the last block of one that ends with the
  // plane (level b below its last pair in the block's last 2 * OFFL iterations); DRAIN: level a below ITS last pair
  auto iter = [&](auto U0c, auto BMc, auto DRc) __attribute__((always_inline)) {
    constexpr int U0 = decltype(U0c)::value, BM = decltype(BMc)::value;
    constexpr bool DRAIN = decltype(DRc)::value;
    if constexpr (!DRAIN || U0 < OFFL) {
      constexpr int UA = U0 & 3;
      constexpr int MODEA = DRAIN ? 2 : (BM == 1 && U0 < 4) ? 1 : 0;
      const int m = DRAIN ? npA + U0 : mb + U0;
      int ll[4];
      if constexpr (!DRAIN && (U0 & 1) == 0) {
        constexpr int J = U0 / 2, UB = (J + OFFL) & 3;
        constexpr int DB = J + OFFL - 4; // (last block) pairs below level b's last
        constexpr int MODEB = (BM == 1 && J + OFFL < 4) ? 1 : (BM == 3 && DB >= 0) ? 2 : 0;
#ifdef VC2_PAIR_NO_B // (timing experiment: level a alone in this kernel's frame)
        ll[0] = ll[1] = ll[2] = ll[3] = 0;
#else
        if (!(dskip & 8)) stepB(IC<UB>(), IC<MODEB>(), IC<(MODEB == 2 ? DB : UB)>(), (mb >> 1) + J + OFFL, true, ll);
        else { ll[0] = ll[1] = ll[2] = ll[3] = 0; }
#endif
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) ll[j] = llO[j];
      }
      RowT<4> re, ro;
      if constexpr (!DRAIN) {
        const int ml = min(m, mloadA), sv = ml >> lbshA;
        if (__builtin_expect(sv != svA_have, 0)) { load_qA(sv); svA_have = sv; }
        int v[12];
        unpackA(ml, U0 % PFQ, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) { re[j] = ll[j]; re[4 + j] = v[j]; ro[j] = v[4 + j]; ro[4 + j] = v[8 + j]; }
      }
      if (pend_k >= 0) { put_out(2 * pend_k, pe); put_out(2 * pend_k + 1, po); }
      if constexpr (!DRAIN) fetchA(min(m + PFQ, mloadA), U0 % PFQ);
      engA.template step<UA, MODEA, (DRAIN ? U0 : UA)>(m, npA, re, ro);
      const int k = m - OFFL;
      if (k >= kA && k < kB) {
        RowT<4> oe = engA.template out<UA>(false), oo = engA.template out<UA>(true);
        make_out(2 * k, oe, pe);
        make_out(2 * k + 1, oo, po);
        pend_k = k;
      } else pend_k = -1;
    }
  };
  auto block = [&](auto BMc, auto DRc) __attribute__((always_inline)) {
    iter(IC<0>(), BMc, DRc); iter(IC<1>(), BMc, DRc); iter(IC<2>(), BMc, DRc); iter(IC<3>(), BMc, DRc);
    iter(IC<4>(), BMc, DRc); iter(IC<5>(), BMc, DRc); iter(IC<6>(), BMc, DRc); iter(IC<7>(), BMc, DRc);
  };
  // level b's run-in: the pairs before the one that enters with level a's first.  A walk from the plane's top: pairs
  // 0 .. OFFL-1 (they fill the rings); any other: OFFL - SD pairs whose results only warm the rings up
  {
    int none[4];
    auto pro = [&](auto Ic, auto TOPc) __attribute__((always_inline)) {
      constexpr int I = decltype(Ic)::value;
      constexpr bool TOP = decltype(TOPc)::value;
      constexpr int NPRO = TOP ? OFFL : OFFL - SD;
      if constexpr (I < NPRO) {
        constexpr int REL = OFFL - NPRO + I;                 // pair index relative to m0 / 2 (negative above it)
        constexpr int UB = ((REL % 4) + 4) % 4;
        const int m1 = (m0 >> 1) + REL;
        fetchB(min(max(m1, 0), mloadB));
        stepB(IC<UB>(), IC<(TOP ? 1 : 0)>(), IC<UB>(), m1, false, none);
      }
    };
    if (top) { pro(IC<0>(), BC<true>()); pro(IC<1>(), BC<true>()); pro(IC<2>(), BC<true>()); pro(IC<3>(), BC<true>()); }
    else {
      pro(IC<0>(), BC<false>()); pro(IC<1>(), BC<false>()); pro(IC<2>(), BC<false>()); pro(IC<3>(), BC<false>());
      pro(IC<4>(), BC<false>()); pro(IC<5>(), BC<false>()); pro(IC<6>(), BC<false>()); pro(IC<7>(), BC<false>());
    }
    fetchB(min((m0 >> 1) + OFFL, mloadB));
  }
#pragma unroll
  for (int u = 0; u < PFQ; ++u) fetchA(min(m0 + u, mloadA), u);
  static_assert(8 % PFQ == 0, "the prefetch ring shares the unrolled walk");
  if (top) {
    block(IC<1>(), BC<false>());
    mb += 8;
  }
  const int prio0 = (int)(blockIdx.x >> 10);
  const int mlast = bottom ? npA - 8 : mend; // (a bottom walk's last block has its own form)
  for (; mb < mlast; mb += 8) {
    if (st_prio) pair_prio(st_prio, prio0, mb >> 3);
    block(IC<0>(), BC<false>());
  }
  if (bottom) {
    if constexpr (OFFL > 0) block(IC<3>(), BC<false>());
    else block(IC<0>(), BC<false>());
    mb += 8;
    block(IC<0>(), BC<true>());
  }
  if (pend_k >= 0) { put_out(2 * pend_k, pe); put_out(2 * pend_k + 1, po); }
#undef p
#undef pb
}

// ------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------
// Which instantiations exist.  The inverse pair that ends with the picture's samples is not used by the library (measured
// slower than its two one-level kernels, vc2hip_api.hip run_inverse): only the tools' build carries it, for A/B runs
// (VC2HIP_PAIR_INV_FINAL=1).  VC2_PAIR_ONLY_*: quick compiles while working on this file.
#ifdef VC2HIP_ABLATE
#define VC2_PAIR_INV_EDGE true
#else
#define VC2_PAIR_INV_EDGE false
#endif
#ifdef VC2_PAIR_ONLY_INV
#define VC2_PAIR_HAS(INV, EDGE) ((INV) && (!(EDGE) || VC2_PAIR_INV_EDGE))
#elif defined(VC2_PAIR_ONLY_FWD)
#define VC2_PAIR_HAS(INV, EDGE) (!(INV))
#else
#define VC2_PAIR_HAS(INV, EDGE) (!(INV) || !(EDGE) || VC2_PAIR_INV_EDGE)
#endif
template <int K, bool EDGE, bool INV, class ST> const void *pair_fn(bool SPL2) {
  if constexpr (!pair_kernel<K>() || !VC2_PAIR_HAS(INV, EDGE)) return nullptr;
  else if constexpr (INV) return SPL2 ? (const void *)k_inv_pair<K, EDGE, ST, true> : (const void *)k_inv_pair<K, EDGE, ST, false>;
  else return (const void *)k_fwd_pair<K, EDGE, ST>;
}
template <int K, bool EDGE, bool INV, class ST>
int launch_pair(Launcher &L, const PairParams &pp, int n_pictures, size_t lds, hipStream_t s) {
  if constexpr (!pair_kernel<K>() || !VC2_PAIR_HAS(INV, EDGE)) return VC2HIP_EINVAL;
  else {
    const LevelParams &p = pp.a;
    const int cols = (p.st_strips[0] + p.st_strips[1] + p.st_strips[2]) * n_pictures;
    const int gx = ((cols + 7) / 8) * p.st_segmax * 8;
    dim3 grid(gx), block(64);
    if constexpr (INV) {
      vc2_prof_begin(L, EDGE ? "idwt_pair_final" : "idwt_pair", s);
      if (pp.spl[0] == 2 || pp.spl[1] == 2 || pp.spl[2] == 2) VC2_LAUNCH(L, (k_inv_pair<K, EDGE, ST, true>), grid, block, 0, s, pp);
      else VC2_LAUNCH(L, (k_inv_pair<K, EDGE, ST, false>), grid, block, 0, s, pp);
    } else {
      vc2_prof_begin(L, EDGE ? "dwt_pair_first" : "dwt_pair", s);
      vc2_allow_lds((const void *)k_fwd_pair<K, EDGE, ST>, std::max<size_t>(64 * 1024, lds));
      VC2_LAUNCH(L, (k_fwd_pair<K, EDGE, ST>), grid, block, lds, s, pp);
    }
    vc2_prof_end(L, s);
    return 0;
  }
}
template <int K, bool EDGE, bool INV, class ST> int pair_slots_of(size_t lds) {
  int nb = 0, dev = 0;
  hipDeviceProp_t prop;
  const void *fn = pair_fn<K, EDGE, INV, ST>(false); // (both forms of the inverse kernel take the same registers)
  if (!fn) return 0;
  if (!INV) vc2_allow_lds(fn, std::max<size_t>(64 * 1024, lds));
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64, INV ? 0 : lds) != hipSuccess || nb < 1)
    return 256 * 8;
  static const int whole = vc2_tune_int("VC2HIP_PAIR_WHOLE_SIMDS", 0);
  if (nb >= 4 && whole) nb &= ~3;
  return nb * prop.multiProcessorCount;
}
#ifdef VC2_PAIR_ONE // (quick compiles while working on this file)
#define VC2_PAIR_CASES(X) X(VC2HIP_DD97)
#else
#define VC2_PAIR_CASES(X) X(VC2HIP_DD97) X(VC2HIP_LEGALL) X(VC2HIP_DD137) X(VC2HIP_HAAR0) X(VC2HIP_HAAR1)
#endif
template <bool INV, class ST> int pair_dispatch(Launcher &L, int kernel, bool edge, const PairParams &pp, int n, size_t lds, hipStream_t s) {
#define VC2_CASE(KK) case KK: return edge ? launch_pair<KK, true, INV, ST>(L, pp, n, lds, s) : launch_pair<KK, false, INV, ST>(L, pp, n, lds, s);
  switch (kernel) { VC2_PAIR_CASES(VC2_CASE) }
#undef VC2_CASE
  return VC2HIP_EINVAL;
}
template <bool INV, class ST> int pair_slots_dispatch(int kernel, bool edge, size_t lds) {
#define VC2_CASE(KK) case KK: return edge ? pair_slots_of<KK, true, INV, ST>(lds) : pair_slots_of<KK, false, INV, ST>(lds);
  switch (kernel) { VC2_PAIR_CASES(VC2_CASE) }
#undef VC2_CASE
  return 0;
}
int pair_slots(int kernel, bool edge, bool inverse, bool store16, size_t lds) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int, size_t, int>, int> cache; // (per device: one process may drive several GPUs)
  int dev = 0;
  (void)hipGetDevice(&dev);
  const auto key = std::make_tuple(kernel, (int)edge, (int)inverse, (int)store16, lds, dev);
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int v;
  if (!store16) v = 0;
  else if (inverse) v = pair_slots_dispatch<true, int16_t>(kernel, edge, lds);
  else v = pair_slots_dispatch<false, int16_t>(kernel, edge, lds);
  cache[key] = v;
  return v;
}
struct PairTaps { int halo, offl, sd; };
template <int K> constexpr PairTaps taps_of(bool inv) {
  return inv ? PairTaps{pair_halo<K>(), VT<K, true>::OFFL, VT<K, true>::sum_dmin()} : PairTaps{pair_halo<K>(), VT<K, false>::OFFL, VT<K, false>::sum_dmin()};
}
bool pair_taps(int kernel, bool inv, PairTaps &t) {
  switch (kernel) {
#define VC2_CASE(KK) case KK: t = taps_of<KK>(inv); return true;
    VC2_PAIR_CASES(VC2_CASE)
#undef VC2_CASE
  }
  return false;
}
bool pow2p(int v) { return v > 0 && (v & (v - 1)) == 0; }
int spl_of(int fw) { return fw >= 8 ? 1 : 2; }
int ilog2h(int v) { return 31 - __builtin_clz((unsigned)v); }

// elements from one slice's run to the next in an LDS image: the run rounded up to whole 16-byte pieces, plus the padding
// (in 16-byte steps) under which the 64 lanes' writes of one block row spread best over the 64 banks
int image_stride(int run_elems, int elem, int lps, int spl, int vals) {
  const int base = (run_elems * elem + 15) & ~15;
  int best = base, best_cost = 1 << 30;
  for (int pad = 0; pad <= 256; pad += 16) {
    const int ss = base + pad;
    int bank[64] = {0};
    for (int l = 0; l < 64; ++l)
      for (int q = 0; q < spl; ++q) {
        const int slice = spl == 1 ? l / lps : l * spl + q;
        const int off = spl == 1 ? (l % lps) * vals * elem : 0;
        const int bytes = std::max(4, vals / spl * elem);
        for (int b = 0; b < bytes; b += 4) ++bank[((slice * ss + off + b) >> 2) & 63];
      }
    int cost = 0;
    for (int b = 0; b < 64; ++b) cost = std::max(cost, bank[b]);
    if (cost < best_cost) { best_cost = cost; best = ss; }
  }
  return best / elem;
}

} // namespace

// Levels a (p.a) and a + 1 (p.b) in one launch?  Fills the st_* fields of pp.a, the image layout, and returns the dynamic
// LDS bytes (0: not applicable -- the caller falls back to one launch per level).
size_t vc2_pair_applicable(PairParams &pp, int kernel, bool edge, bool inverse, bool store16, int n_pictures) {
  PairTaps t;
  if (!pair_taps(kernel, inverse, t)) return 0;
  LevelParams &p = pp.a;
  const LevelParams &pb = pp.b;
  const int elem = store16 ? 2 : 4;
  const bool b_last = inverse ? pb.ll_from_store != 0 : pb.ll_to_store != 0;
  if (inverse && (p.bp8 || pb.bp8)) return 0; // (byte band planes: read by the one-level kernels only)
  if (!store16) return 0; // (the int32 store -- fine-grained calls, the STORE32 test variant -- keeps one launch per level: half of this file's compile time)
  size_t imgA = 16, imgB = 16;
  p.st_tail = 0;
  p.st_segmax = 0;
  for (int c = 0; c < 3; ++c) {
    p.st_strips[c] = p.st_segs[c] = 0;
    pp.spl[c] = 1;
    if (p.tiles_x[c] == 0 || p.tiles_y[c] == 0) continue;
    const int w = p.in_w[c], h = p.in_h[c], fw = p.fw[c], fh = p.fh[c];
    if (w < VC2_STREAM_MIN_W || (w & 7) || (h & 3) || h < 48) return 0;
    if (edge && (p.word_bytes != 2 || p.pic_w[c] != w)) return 0;
    if (!pow2p(fw) || !pow2p(fh) || fw < 4 || fh < 4 || fw > 64 * 8) return 0;
    if (pb.fw[c] * 2 != fw || pb.fh[c] * 2 != fh || pb.in_w[c] * 2 != w || pb.in_h[c] * 2 != h) return 0;
    const int spl = fw >= 8 ? 1 : 2, lps = fw >= 8 ? fw / 8 : 1;
    const int bnA = (fh / 2) * (fw / 2), bnB = (fh / 4) * (fw / 4);
    const int runA = 3 * bnA, runB = (b_last ? 4 : 3) * bnB;
    // a run moves in pieces of 16 bytes, or of 8 / 4 where its bands are shorter (a band at the levels between, the
    // whole [LL | HL | LH | HH] at the last level starts on a multiple of its own length)
    const int pieceA = std::min(16, bnA * elem), pieceB = std::min(16, (b_last ? 4 * bnB : bnB) * elem);
    if (!inverse) {
      if (pieceA < 4 || pieceB < 4) return 0;
      if ((p.coef_off[c] * elem) % 16 || (pb.coef_off[c] * elem) % 16 || (p.slice_coefs * elem) % 16) return 0;
    } else { // the inverse kernel's loads: four (two where the lane spans two slices) coefficients of level a, two (one) of level b
      if ((p.coef_off[c] | pb.coef_off[c] | p.rec_stride[c] | pb.rec_stride[c] | bnA) & 3) return 0;
      if ((bnB & 1) && !(bnB == 1 && b_last && spl_of(fw) == 2)) return 0;
    }
    pp.piece_a[c] = pieceA;
    pp.piece_b[c] = pieceB;
    pp.spl[c] = spl;
    static const int out_cap = vc2_tune_int("VC2HIP_PAIR_OUT", 64);
    const int out = (std::min(64 - 2 * t.halo, out_cap) / lps) * lps;
    if (out < lps) return 0;
    const int nch = w / 8;
    p.st_out[c] = out;
    p.st_llps[c] = ilog2h(lps);
    p.st_strips[c] = (nch + out - 1) / out;
    const int nsl = out / lps * spl;
    pp.ss_a[c] = image_stride(runA, elem, lps, spl, 4);
    pp.ss_b[c] = (runB * elem + 15) / 16 * 16 / elem; // (unpadded: the two images of UHD then fit eleven times into a CU's LDS)
    imgA = std::max(imgA, (size_t)nsl * pp.ss_a[c] * elem);
    imgB = std::max(imgB, (size_t)nsl * pp.ss_b[c] * elem);
  }
  if (inverse) imgA = imgB = 0;
  imgA = (imgA + 15) & ~(size_t)15;
  imgB = (imgB + 15) & ~(size_t)15;
  // Slice rows per burst of stores: a burst drains the wavefront's memory operations, so short slices (the deep levels:
  // two row pairs per slice row) are stored in groups -- as many slice rows as make eight row pairs, while the images of
  // eight wavefronts per CU fit its LDS
  int bsh_min = 1 << 20, bshb_max = 1;
  for (int c = 0; c < 3; ++c) if (p.st_strips[c]) { bsh_min = std::min(bsh_min, p.fh[c] / 2); bshb_max = std::max(bshb_max, p.fh[c] / 4); }
  int grp = 1;
  auto lds_of = [&](int g) {
    int nll = 4;
    while (nll < g * bshb_max) nll *= 2;
    return (size_t)g * imgA + (size_t)(g + 1) * imgB + (b_last ? 0 : (size_t)nll * 64 * (store16 ? 4 : 8));
  };
  while (grp * 2 * bsh_min <= 8 && lds_of(grp * 2) <= 20 * 1024) grp *= 2;
  static const int force_grp = vc2_tune_int("VC2HIP_PAIR_GROUP", 0);
  if (force_grp > 0) grp = force_grp;
  int nll = 4;
  while (nll < grp * bshb_max) nll *= 2;
  pp.grp_a = grp;
  pp.sz_a = (int)(imgA / elem);
  pp.sz_b = (int)(imgB / elem);
  pp.img_b = (int)(grp * imgA);
  pp.ring_ll = (int)(grp * imgA + (grp + 1) * imgB);
  pp.n_ll = nll;
  const size_t lds = inverse ? 16 : lds_of(grp);
  if (lds > 48 * 1024) return 0;
  // Segments as in the one-level kernels (whole rows of slices, the count that minimises rounds x rows walked), with the
  // conditions of this kernel's unrolled walk: a strip with one segment needs a pair count that is a multiple of eight
  // (its walk starts at the top AND ends at the bottom); with more, the last segment must lie wholly below the rows the
  // one above it reads, and start far enough down that its walk can begin a whole number of blocks above the plane's end
  int cols = 0;
  for (int c = 0; c < 3; ++c) if (p.st_strips[c]) cols += p.st_strips[c] * n_pictures;
  const int slots = pair_slots(kernel, edge, inverse, store16, lds);
  if (slots <= 0) return 0;
  const int runin = -3 * t.sd + 3 * t.offl + 8;
  int nseg = 0;
  long long best = -1;
  for (int g = 1; g <= p.ys; ++g) {
    bool ok = true;
    int tallest = 0;
    for (int c = 0; c < 3 && ok; ++c) {
      if (!p.st_strips[c]) continue;
      const int bsh = p.fh[c] / 2, np = p.in_h[c] / 2;
      tallest = std::max(tallest, ((p.ys + g - 1) / g) * bsh);
      if (g == 1) { ok = np % 8 == 0; continue; }
      const int ka_bottom = (int)((long long)(g - 1) * p.ys / g) * bsh;
      if (np - ka_bottom < 3 * t.offl) ok = false;            // the segment above reads 3 * OFFL pairs below its own
      if (ka_bottom + 3 * t.sd - 8 < 0) ok = false;           // room for the bottom walk's start
    }
    if (!ok) continue;
    const long long rounds = ((long long)cols * g + slots - 1) / slots;
    const long long cost = rounds * ((long long)tallest + runin);
    if (best < 0 || cost <= best) { best = cost; nseg = g; } // (ties: more wavefronts)
  }
  if (!nseg) return 0;
  {
    static const int f = vc2_tune_int("VC2HIP_PAIR_NSEG", 0), fe = vc2_tune_int("VC2HIP_PAIR_NSEG_EDGE", 0);
    if (f > 0) nseg = std::min(f, p.ys);
    if (fe > 0 && edge) nseg = std::min(fe, p.ys);
  }
  for (int c = 0; c < 3; ++c) if (p.st_strips[c]) p.st_segs[c] = nseg;
  p.st_segmax = nseg;
  p.st_npic = n_pictures;
  static const int prio = vc2_tune_int("VC2HIP_PAIR_PRIO", 1); // (measured: rotation 0.552 - 0.561, alternation 0.542 - 0.558, longer turns 0.549 - 0.569, none 0.574 - 0.583 ms)
  p.st_prio = prio; // (wavefronts take turns at the highest issue priority, a new turn every block: as the one-level kernels)
  p.st_lds = (int)lds;
#ifdef VC2HIP_ABLATE
  if (getenv("VC2HIP_PAIR_DEBUG")) fprintf(stderr, "pair: kernel %d edge %d lds %zu (img a %zu) slots %d cols %d nseg %d runin %d\n", kernel, (int)edge, lds, imgA, slots, cols, nseg, runin);
#endif
  return lds;
}
int vc2_launch_forward_pair(Launcher &L, int kernel, bool first, const PairParams &pp, int n, bool store16, size_t lds, hipStream_t s) {
  return store16 ? pair_dispatch<false, int16_t>(L, kernel, first, pp, n, lds, s) : VC2HIP_EINVAL;
}
int vc2_launch_inverse_pair(Launcher &L, int kernel, bool final_level, const PairParams &pp, int n, bool store16, size_t lds, hipStream_t s) {
  return store16 ? pair_dispatch<true, int16_t>(L, kernel, final_level, pp, n, lds, s) : VC2HIP_EINVAL;
}
