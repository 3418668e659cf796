// Streaming DWT / IDWT level kernels: the hot configuration (planes from 192 samples wide, slices of whole 8-sample
// chunks; all seven wavelets).
//
// One wavefront owns a strip of 64 chunks (8 samples each: one 16-byte load per lane and row, 1 KiB per wavefront) and
// walks down the rows of its segment.  Nothing is shared between wavefronts and there is no workgroup barrier:
//   * horizontal lifting runs in registers on the lane's 4 coefficient pairs; the taps beyond the chunk come from the
//     neighbouring lanes with DPP wavefront shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1).  The first / last lane of
//     a wavefront has no such neighbour and keeps the `old` operand, which is set to the lane's own edge pair: when
//     the strip starts / ends at the plane edge that IS the reference's tap clamping (WaveletTransform.cpp:478-1265:
//     even taps clamp to [0, n-2], odd taps to [1, n-1]); inside the plane those lanes are halo lanes whose results
//     are not written (strips overlap by the halo).  A plane narrower than the wavefront ends at an inner lane, which
//     clamps by a select; the lanes behind it idle.
//   * vertical lifting is a line-based scheme: the rows still needed by a later lifting step stay in registers (rings
//     of 4 rows per sequence, 8 for Fidelity's 8-tap steps); every new row pair completes one output row pair a few
//     rows higher up.  Rows above / below the plane replicate the first / last pair (the same clamping).
//   * rows are prefetched ahead into registers; every iteration's stores are issued at the top of the next one.
//   * forward: the coefficient store keeps every slice's coefficients together, so the three detail bands
//     pass through a small wavefront-private LDS image laid out [band][row][slice][column] and move to the slice
//     records as whole 16-byte pieces of the contiguous [HL | LH | HH] run of a slice, one block row of slices at a time.
//   * inverse: four coefficients per band and lane straight from the decoder's band planes (BandPlanes,
//     vc2hip_internal.h) or, for levels that have none, from the slice records.
// Same LevelParams and the same results as the tile kernels of vc2hip_dwt_fast.hip, which remain for every geometry
// this scheme does not cover (slice footprints below one chunk, very narrow planes, padded widths, Fidelity planes
// that are not whole blocks of eight row pairs).
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

__constant__ QuantTables c_qst;
void vc2_upload_tables_stream(const QuantTables &t, hipStream_t s) {
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(c_qst), &t, sizeof t, 0, hipMemcpyHostToDevice, s);
}

#ifdef VC2HIP_STAMPS // diagnostic build only (tools/stamps.py): per-wavefront start / end times of the streaming kernels
__device__ unsigned long long *g_stamps;
#define VC2_STAMP_BEGIN const unsigned long long stamp_t0 = wall_clock64();
#define VC2_STAMP_END(work) do { if (threadIdx.x == 0 && g_stamps) { \
    const size_t bid = blockIdx.x; \
    unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
    g_stamps[4 * bid] = stamp_t0; g_stamps[4 * bid + 1] = wall_clock64(); g_stamps[4 * bid + 2] = ((unsigned long long)xcc << 32) | hw; g_stamps[4 * bid + 3] = (work); } } while (0)
#else
#define VC2_STAMP_BEGIN
#define VC2_STAMP_END(work)
#endif

namespace {

#include "vc2hip_stream_eng.h"

// ------------------------------------------------------------------------------------------
// forward level
// ------------------------------------------------------------------------------------------
// TAIL: the plane's pair count need not be a multiple of the ring length (its own instantiation, levels below the first
// only: the tails cost registers -- 154 instead of 113 -- and the level-0 kernels of every BASELINE format do not need them)
// wavefronts per SIMD the register allocator must leave room for: the steady instantiations live on occupancy (their
// budgets -- 128 registers, 256 for Fidelity's rings of eight -- are met without scratch; a few registers more would halve
// or quarter the wavefronts), the TAIL ones (rarely used, small planes) take what they need
#ifndef VC2_STREAM_WPE
#define VC2_STREAM_WPE 4
#endif
template <int K, bool TAIL> constexpr int stream_wpe() { return TAIL ? 1 : K == VC2HIP_FIDELITY ? 2 : VC2_STREAM_WPE; }
template <int K, bool FIRST, class ST, bool TAIL = false>
__global__ __launch_bounds__(64 * VC2_STREAM_WG_WAVES, (stream_wpe<K, TAIL>())) void k_fwd_stream(const LevelParams p) {
  using S_ = St<ST>;
  using VE = VEng<K, false>;
  using T = typename VE::T;
  constexpr int RL = RLK<K>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  ST *stg = (ST *)(smem + (size_t)stream_wave() * p.st_lds);
  const int lane = threadIdx.x & 63;
  int comp, pic;
  VC2_STAMP_BEGIN
  Strip sp;
  if (!strip_of_block<K>(p, comp, pic, sp)) { VC2_STAMP_END(0); return; }
  constexpr int ACC = WT<K>::accuracy;
  const int in_h = p.in_h[comp], in_w = p.in_w[comp], np = in_h >> 1;
  const int chunk = min(sp.c0 + lane, (in_w >> 3) - 1); // (lanes behind a narrow plane repeat its last chunk and own nothing)
  const bool own = lane >= sp.lo && lane < sp.hi;
  const bool redge = sp.c0 + lane == (in_w >> 3) - 1 && lane != 63;

  // ---- input rows
  const uint8_t *raw = nullptr;
  const ST *lvl = nullptr;
  const int32_t *lvl_w = nullptr;
  if constexpr (FIRST) raw = (const uint8_t *)p.plane[comp] + (size_t)pic * p.plane_stride[comp] + (size_t)chunk * 16;
  else {
    lvl = (const ST *)p.plane[comp] + (size_t)pic * p.plane_stride[comp] + (size_t)chunk * 8;
    if constexpr (S_::narrow) lvl_w = p.plane_wide[comp] + (size_t)pic * p.plane_stride[comp] + (size_t)chunk * 8;
  }
  const int pic_h = FIRST ? p.pic_h[comp] : in_h;
  constexpr int NQ = (FIRST || S_::narrow) ? 1 : 2; // 16-byte loads per row
  uint4 pf[PF][2][NQ];
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
  auto convert = [&](int m, int slot, int h, Row &r) __attribute__((always_inline)) {
    if constexpr (FIRST) {
      const unsigned w[4] = {pf[slot][h][0].x, pf[slot][h][0].y, pf[slot][h][0].z, pf[slot][h][0].w};
      const int sshift = comp ? p.sample_shift_c : p.sample_shift; // (wave-uniform: chroma words may have their own depth)
      const int bias = -((comp ? p.sample_offset_c : p.sample_offset) << ACC);
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

  // ---- outputs
  const int fh = p.fh[comp], fw = p.fw[comp];
  const int bsh = fh >> 1, bsw = fw >> 1, lbsh = ilog2d(bsh), lbsw = ilog2d(bsw), bn = bsh * bsw;
  const int nb = p.ll_to_store ? 4 : 3, b0 = p.ll_to_store ? 0 : 1; // bands through the LDS image; first of them
  const int rs = p.st_out[comp] * 4;                                 // elements per image row (all owned slices)
  const int si = (lane - sp.lo) >> p.st_llps[comp];                  // slice of the lane inside the strip
  const int cc = ((lane - sp.lo) & ((1 << p.st_llps[comp]) - 1)) * 4; // its first column inside the slice's block
  const int run0 = p.coef_off[comp] + (p.ll_to_store ? 0 : bn);      // [LL |] HL | LH | HH of a slice: one contiguous run
  ST *store = (ST *)p.store + (size_t)pic * p.store_stride;
  int32_t *wide = S_::narrow ? p.store_wide + (size_t)pic * p.store_stride : nullptr;
  ST *llp = nullptr;
  int32_t *llp_w = nullptr;
  if (!p.ll_to_store) {
    llp = (ST *)p.ll[comp] + (size_t)pic * p.ll_stride[comp] + (size_t)chunk * 4;
    if constexpr (S_::narrow) llp_w = p.ll_wide[comp] + (size_t)pic * p.ll_stride[comp] + (size_t)chunk * 4;
  }
  const int ow = in_w >> 1;

  // four band values of the lane into the image row (band b, block row r); values beyond 16 bits go to the wide plane
  auto stage4 = [&](int b, int r, int sv, int a0, int a1, int a2, int a3) __attribute__((always_inline)) {
#ifdef VC2_STREAM_DIRECT // experiment: the lane's four values straight into the slice record (8 / 16 bytes per lane), no LDS image, no flush
    ST *d = store + mul24z(sv * p.xs + sp.sx0 + si, p.slice_coefs) + run0 + (b - b0) * bn + r * bsw + cc;
#else
    ST *d = stg + ((size_t)((b - b0) * bsh + r) * rs + si * bsw + cc);
#endif
    if constexpr (S_::narrow) {
      const int mx = max(max(a0, a1), max(a2, a3)), mn = min(min(a0, a1), min(a2, a3));
      if (mx > 32767 || mn < -32767) {
        int32_t *w = wide + (size_t)(sv * p.xs + sp.sx0 + si) * p.slice_coefs + run0 + (b - b0) * bn + r * bsw + cc;
        if (!S_::fits(a0)) { w[0] = a0; a0 = VC2_ST_SENTINEL; }
        if (!S_::fits(a1)) { w[1] = a1; a1 = VC2_ST_SENTINEL; }
        if (!S_::fits(a2)) { w[2] = a2; a2 = VC2_ST_SENTINEL; }
        if (!S_::fits(a3)) { w[3] = a3; a3 = VC2_ST_SENTINEL; }
      }
      *(uint2 *)d = make_uint2(vc2_pack16(a0, a1), vc2_pack16(a2, a3));
    } else *(int4 *)d = make_int4(a0, a1, a2, a3);
  };
  // the image of one block row of slices to the slice records: 16-byte pieces of each slice's contiguous run
  auto flush = [&](int sv) __attribute__((always_inline)) {
#ifdef VC2_STREAM_DIRECT
    return;
#endif
    wave_sync();
    constexpr int EP = 16 / (int)sizeof(ST);             // elements per piece
    const int lpb = ilog2d(bn) - ilog2d(EP);             // log2 pieces per band block (bn >= EP: host check)
    for (int b = 0; b < nb; ++b) {
      for (int q = lane; q < (sp.nsl << lpb); q += 64) {
        const int s2 = q >> lpb, e = (q & ((1 << lpb) - 1)) * EP; // slice, first element inside the band block
        const int r = e >> lbsw, c = e & (bsw - 1);
        const size_t at = mul24z(sv * p.xs + sp.sx0 + s2, p.slice_coefs) + run0 + b * bn + e;
        const ST *src = stg + ((size_t)(b * bsh + r) * rs + s2 * bsw + c);
        if (bsw >= EP) *(uint4 *)(store + at) = *(const uint4 *)src;
        else { // a piece spans two block rows (bsw == EP / 2)
          const uint2 lo = *(const uint2 *)src, hi = *(const uint2 *)(src + rs);
          *(uint4 *)(store + at) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      }
    }
    wave_sync();
  };

  // ---- the walk
  VE eng;
  eng.clear();
  // Every iteration issues all its global memory operations in one place, right after the rows it consumes have
  // arrived: the stores of the PREVIOUS iteration's results (the LL row; the LDS image when a block row is complete)
  // and the load of the pair PF ahead.  The compiler's wait for the consumed rows (a wait for everything outstanding:
  // the waits inside this loop are not counted ones) then finds operations that had a whole iteration to complete.
  const int m0 = max(sp.kA + T::sum_dmin(), 0) & ~(RL - 1); // walks start at multiples of RL (a longer run-in is harmless)
  // the walk reaches the plane's bottom (its steady blocks would run past the last pair): the bottom handling, whether or
  // not the segment's own rows end there (rows beyond kB are computed and not stored)
  const bool last = sp.kB + T::OFFL + RL - 1 > np;
  const int mend = last ? np : sp.kB + T::OFFL;             // steady-state iterations: [m0, mend) in whole blocks of RL
  const int mload = np - 1;                                 // loads beyond it repeat the last pair (their results are not used)
#pragma unroll
  for (int u = 0; u < PF; ++u) fetch(min(m0 + u, mload), u);
  static_assert(RL % PF == 0, "the prefetch ring shares the unrolled walk of the row rings");
  int ll_k = -1, flush_sv = -1; // deferred stores: LL row of pair ll_k (values in llv), image of block row flush_sv
  int llv[4] = {0, 0, 0, 0};
#define VC2_FWD_ITER(U, MODE) VC2_FWD_ITERD(U, MODE, U)
#define VC2_FWD_ITERD(U, MODE, D)                                                                            \
  if constexpr (MODE != 2 || D < T::OFFL) {                                                                  \
    const int m = MODE == 2 ? np + D : mb + U;                                                               \
    Row re, ro;                                                                                              \
    if constexpr (MODE != 2) {                                                                               \
      convert(min(m, mload), U % PF, 0, re);                                                                 \
      convert(min(m, mload), U % PF, 1, ro);                                                                 \
    }                                                                                                        \
    if (ll_k >= 0 && own) S_::store4(llp + mul24z(ll_k, ow), llp_w + mul24z(ll_k, ow), llv[0], llv[1], llv[2], llv[3]); \
    if (flush_sv >= 0) { flush(flush_sv); flush_sv = -1; }                                                   \
    if constexpr (MODE != 2) {                                                                               \
      fetch(min(m + PF, mload), U % PF);                                                                     \
      h_lift<K, false>(re, redge);                                                                           \
      h_lift<K, false>(ro, redge);                                                                           \
    }                                                                                                        \
    eng.template step<U, MODE, D>(m, np, re, ro);                                                            \
    const int k = m - T::OFFL;                                                                               \
    if (k >= sp.kA && k < sp.kB) {                                                                           \
      const Row &oe = eng.template out<U>(false), &oo = eng.template out<U>(true);                           \
      const int r = k & (bsh - 1), sv = k >> lbsh;                                                           \
      if (p.ll_to_store) { if (own) stage4(0, r, sv, oe[0], oe[1], oe[2], oe[3]); }                          \
      else { ll_k = k; llv[0] = oe[0]; llv[1] = oe[1]; llv[2] = oe[2]; llv[3] = oe[3]; }                     \
      if (own) {                                                                                             \
        stage4(1, r, sv, oe[4], oe[5], oe[6], oe[7]);                                                        \
        stage4(2, r, sv, oo[0], oo[1], oo[2], oo[3]);                                                        \
        stage4(3, r, sv, oo[4], oo[5], oo[6], oo[7]);                                                        \
      }                                                                                                      \
      if (r == bsh - 1) flush_sv = sv;                                                                       \
    } else ll_k = -1;                                                                                        \
  }
#define VC2_FWD_BLOCK(MODE) { VC2_FWD_ITER(0, MODE) VC2_FWD_ITER(1, MODE) VC2_FWD_ITER(2, MODE) VC2_FWD_ITER(3, MODE) \
    if constexpr (RL == 8) { VC2_FWD_ITER(4, MODE) VC2_FWD_ITER(5, MODE) VC2_FWD_ITER(6, MODE) VC2_FWD_ITER(7, MODE) } }
  int mb = m0;
  if (m0 == 0) { // (the host admits planes of at least 2 * RL row pairs: this block lies inside the plane)
    VC2_FWD_BLOCK(1)
    mb += RL;
  }
  const int mstop = last ? (np & ~(RL - 1)) : mend; // whole steady blocks (a walk that does not end at the bottom may run over)
  const int prio0 = VC2_STREAM_WG_WAVES > 1 ? (int)(blockIdx.x >> 8) : (int)(blockIdx.x >> 10); // (the wavefronts that share a SIMD: a chip full of workgroups apart)
  for (; mb < mstop; mb += RL) { if (p.st_prio) prio_turn(prio0 + (mb >> p.st_prio)); VC2_FWD_BLOCK(0) }
  if (last) { // the np mod RL pairs left, then the OFFL iterations below the plane at the phases that follow
#define VC2_FWD_DRAIN(R) { VC2_FWD_ITERD((R + 0) % RL, 2, 0) VC2_FWD_ITERD((R + 1) % RL, 2, 1) VC2_FWD_ITERD((R + 2) % RL, 2, 2) \
    VC2_FWD_ITERD((R + 3) % RL, 2, 3) VC2_FWD_ITERD((R + 4) % RL, 2, 4) VC2_FWD_ITERD((R + 5) % RL, 2, 5) VC2_FWD_ITERD((R + 6) % RL, 2, 6) }
    if constexpr (RL == 4 && TAIL) {
      switch (np & 3) {
        case 0: VC2_FWD_DRAIN(0) break;
        case 1: VC2_FWD_ITER(0, 0) VC2_FWD_DRAIN(1) break;
        case 2: VC2_FWD_ITER(0, 0) VC2_FWD_ITER(1, 0) VC2_FWD_DRAIN(2) break;
        default: VC2_FWD_ITER(0, 0) VC2_FWD_ITER(1, 0) VC2_FWD_ITER(2, 0) VC2_FWD_DRAIN(3) break;
      }
    } else VC2_FWD_DRAIN(0) // (no TAIL, and rings of eight -- eight tails would not fit the instruction cache: the host admits whole blocks only)
#undef VC2_FWD_DRAIN
  }
#undef VC2_FWD_BLOCK
#undef VC2_FWD_ITER
#undef VC2_FWD_ITERD
  if (ll_k >= 0 && own) S_::store4(llp + mul24z(ll_k, ow), llp_w + mul24z(ll_k, ow), llv[0], llv[1], llv[2], llv[3]);
  if (flush_sv >= 0) flush(flush_sv);
  VC2_STAMP_END(1);
}

// ------------------------------------------------------------------------------------------
// inverse level
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int dequant_full(int v, int qf, int off) { // scale(), Quantisation.cpp:86-95, literally
  if (v == 0) return 0;
  const unsigned mag = v < 0 ? 0u - (unsigned)v : (unsigned)v;
  int a = (int)(mag * (unsigned)qf);
  if (a > 0) a = (int)((unsigned)a + (unsigned)off);
  a = (int)((unsigned)a + 2u);
  a /= 4;
  return v < 0 ? (int)(0u - (unsigned)a) : a;
}

#ifndef VC2_STREAM_NT_OUT
#define VC2_STREAM_NT_OUT 1 // the decoded picture's rows leave with non-temporal stores (k_inv_stream FINAL)
#endif
#ifndef VC2_STREAM_DQ8
#define VC2_STREAM_DQ8 2 // byte planes: dequantise through a table per band in LDS (k_inv_stream); 1: the last level only
#endif
// BP8 (round 5): the level's band planes hold ONE BYTE per coefficient (vc2hip_internal.h BandPlanes::bytes8: quantised
// coefficients are small; -128 is the sentinel, the value then sits in the wide array at the element's index as for the
// 16-bit sentinel).  This kernel runs at the memory system's pace: its band rows are half of what it reads.
template <int K, bool FINAL, class ST, bool TAIL = false, bool BP8 = false>
__global__ __launch_bounds__(64 * VC2_STREAM_WG_WAVES, (stream_wpe<K, TAIL>())) void k_inv_stream(const LevelParams p) {
  using S_ = St<ST>;
  static_assert(!BP8 || (S_::narrow && !TAIL), "byte planes: with the 16-bit store, whole blocks");
  using VE = VEng<K, true>;
  using T = typename VE::T;
  constexpr int RL = RLK<K>;
  __shared__ int qtab[360]; // quant_factor / quant_offset / domain limit by adjusted index (every wavefront of the workgroup writes all of it: the same values)
  // BP8: the dequantised value of every byte, per band, for the quantiser index all of the wavefront's slices share (the
  // common case: HQ_ConstQ streams) -- a byte then costs one shift and one LDS read where extraction, range test and
  // scale() cost seven VALU instructions, and this kernel's time is its VALU instructions.  Private to the
  // wavefront; rebuilt when the index changes; slices with different indices side by side take the arithmetic.
  // Measured (same box, alternating): the last inverse level of 32 UHD pictures 0.478 -> 0.418 ms, of 32 HD pictures (LeGall)
  // 0.147 -> 0.109.  The last level of the short filters only: the instantiation for the levels below it has no registers
  // to spare (128 with 40 bytes of scratch, 0.132 -> 0.148 ms with the table), DD137 / Daub97 / Fidelity would spill
  // more than they do (Fidelity, 4 UHD-2 pictures: 0.656 -> 0.667).
  constexpr bool DQ8 = BP8 && (FINAL || VC2_STREAM_DQ8 > 1) && VC2_STREAM_DQ8 &&
                       (K == VC2HIP_DD97 || K == VC2HIP_LEGALL || K == VC2HIP_HAAR0 || K == VC2HIP_HAAR1);
  __shared__ int dq8[DQ8 ? VC2_STREAM_WG_WAVES * 768 : 1];
  int *const tab8 = dq8 + (DQ8 ? (int)(threadIdx.x >> 6) * 768 : 0);
  int tab_q = -1;      // the index the table holds (wave-uniform)
  bool tab_on = false; // the current slice row reads through the table (wave-uniform)
  const int lane = threadIdx.x & 63;
  int comp, pic;
  VC2_STAMP_BEGIN
  Strip sp;
  if (!strip_of_block<K>(p, comp, pic, sp)) { VC2_STAMP_END(0); return; }
  for (int i = lane; i < 120; i += 64) {
    const int qf = c_qst.qf[i], off = c_qst.off[i];
    qtab[i] = qf; qtab[120 + i] = off;
    // largest magnitude for which the five-instruction form below is exact: |v| and the factor inside 23 / 24 bits (the
    // full-rate 24-bit multiplies) and the result (|v| * factor + offset + 2) >> 2 inside 23 bits; beyond it the literal
    // arithmetic
    qtab[240 + i] = (qf > 0 && qf < (1 << 24)) ? (int)min(((1u << 25) - (unsigned)off - 8u) / (unsigned)qf, 0x7FFFFFu) : -1;
  }
  wave_sync();
  constexpr int ACC = WT<K>::accuracy;
  const int out_h = p.in_h[comp], out_w = p.in_w[comp], np = out_h >> 1, ow = out_w >> 1;
  const int chunk = min(sp.c0 + lane, (out_w >> 3) - 1); // (lanes behind a narrow plane repeat its last chunk and own nothing)
  const bool own = lane >= sp.lo && lane < sp.hi;
  const bool redge = sp.c0 + lane == (out_w >> 3) - 1 && lane != 63;
  const int fh = p.fh[comp], fw = p.fw[comp];
  const int bsh = fh >> 1, bsw = fw >> 1, lbsh = ilog2d(bsh), bn = bsh * bsw;
  const int b0 = p.ll_from_store ? 0 : 1;               // first band that comes from the store
  const int llps = p.st_llps[comp];
  const int sx = chunk >> llps, cc = (chunk & ((1 << llps) - 1)) * 4; // the lane's slice and first column of its block
  const int run0 = p.coef_off[comp] + (p.ll_from_store ? 0 : bn);     // [LL |] HL | LH | HH of a slice
  const ST *store = (const ST *)p.store + (size_t)pic * p.store_stride;
  const int32_t *wide = S_::narrow ? p.store_wide + (size_t)pic * p.store_stride : nullptr;
  const int32_t *qidx = p.qidx ? p.qidx + (size_t)pic * p.ys * p.xs : nullptr;
  const ST *llp = nullptr;
  const int32_t *llp_w = nullptr;
  if (!p.ll_from_store) {
    llp = (const ST *)p.ll[comp] + (size_t)pic * p.ll_stride[comp] + (size_t)chunk * 4;
    if constexpr (S_::narrow) llp_w = p.ll_wide[comp] + (size_t)pic * p.ll_stride[comp] + (size_t)chunk * 4;
  }

  // ---- input: per band row the lane's four coefficients of every band, straight from the store, prefetched PFI rows ahead.
  // (A version that brought the bands in through LDS-DMA in whole 128-byte lines, double buffered, was correct but
  // slower -- 0.75 against 0.55 ms per 16 UHD pictures: its 17 KiB image per wavefront halves the occupancy, and this
  // kernel lives on occupancy.)
  typedef typename std::conditional<S_::narrow, uint2, uint4>::type Q4; // four store elements
  Q4 bq[PFI][BP8 ? 1 : 4];
  unsigned bq8[PFI][BP8 ? 4 : 1]; // (BP8) the four bytes of bands 1..3
  // element index of the lane's four coefficients of band b in band row m: in the slice records (a slice's band block
  // row is bsw coefficients: neighbouring lanes read neighbouring 8 / 16 bytes of it, then the next slice's record), or,
  // when the decoder laid this level's bands out as planes (BandPlanes), simply row m, column 4 * chunk of plane b
  const long long bp = p.bp_base[comp];
  auto rec_at = [&](int m, int b) __attribute__((always_inline)) -> size_t {
    if (bp >= 0 && b > 0) return (size_t)bp + (size_t)mul24z((b - 1) * np + m, ow) + (size_t)chunk * 4;
    const int sv = m >> lbsh, r = m & (bsh - 1);
    return mul24z(sv * p.xs + sx, p.slice_coefs) + run0 + (b - b0) * bn + r * bsw + cc;
  };
  auto in_fetch = [&](int m, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b == 0 && !p.ll_from_store) bq[slot][0] = *(const Q4 *)(llp + mul24z(m, ow));
      else if constexpr (BP8) { // byte address of element e of a plane that starts at element bp: 2 bp + (e - bp)
        if (b > 0) bq8[slot][b] = *(const unsigned *)((const char *)store + rec_at(m, b) + (size_t)bp);
      } else bq[slot][b] = *(const Q4 *)(store + rec_at(m, b));
    }
  };
  // quantiser constants of the lane's slice in block row sv, per band
  // (DQ8: the twelve constants are not kept across the walk -- only the four indices in one register; whoever needs the
  // constants fetches them from qtab at that moment (q_now): eleven registers for the table path's addresses)
  int qf[4], qo[4], ql[4];
  unsigned aqp = 0;
  auto q_now = [&]() __attribute__((always_inline)) {
    if constexpr (DQ8) {
#pragma unroll
      for (int b = 0; b < 4; ++b) { const int a = (int)((aqp >> (8 * b)) & 0xFFu); qf[b] = qtab[a]; qo[b] = qtab[120 + a]; ql[b] = qtab[240 + a]; }
    }
  };
  auto load_q = [&](int sv) __attribute__((always_inline)) {
    const int q = p.dequant ? qidx[sv * p.xs + sx] : 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int qm = b == 0 ? p.qmatrix[0] : p.qmatrix[p.band + b - 1];
      const int aq = max(q - qm, 0);
      if (aq > 119 && p.dequant && b >= b0) atomicOr(p.err, VC2_DEVERR_QINDEX);
      if constexpr (DQ8) aqp = b == 0 ? (unsigned)min(aq, 119) : aqp | (unsigned)min(aq, 119) << (8 * b);
      else { qf[b] = qtab[min(aq, 119)]; qo[b] = qtab[120 + min(aq, 119)]; ql[b] = qtab[240 + min(aq, 119)]; }
    }
    if constexpr (DQ8) {
      tab_on = false;
      if (p.dequant && !p.ll_from_store) {
        const int q0 = __builtin_amdgcn_readfirstlane(q);
        if (__builtin_amdgcn_ballot_w64(q != q0) == 0ull) { // one index for every slice of the wavefront's strip
          if (q0 != tab_q) {
            q_now();
            wave_sync(); // (the last row's reads of the old table are done)
#pragma unroll
            // The marker 0x80000000 is no value of scale(): dequant_full ends in `a /= 4` on an int, so every result --
            // the wrapped factors of indices 116..119 included -- lies within +-2^29.  A prefetched row holds raw bytes
            // only and is dequantised where it is used, behind the load_q of ITS slice row: no row ever reads a table
            // built for another row's index (tests/test_gpu_wide.py::test_byte_planes_high_and_mixed_indices).
            for (int i = 0; i < 12; ++i) { // entry 256 (b - 1) + byte, b = 1 + i / 4 (the factors are the same in every lane)
              const int x = (int)(signed char)(64 * (i & 3) + lane);
              tab8[64 * i + lane] = x == -128 ? (int)0x80000000 : dequant_full(x, qf[1 + i / 4], qo[1 + i / 4]);
            }
            wave_sync();
            tab_q = q0;
          }
          tab_on = true;
        }
      }
    }
  };
  // unpack band b of row m (escapes resolved) and dequantise
  // unpack band b of row m (escapes resolved) and dequantise: band by band (the Fidelity kernels)
  auto band4 = [&](int m, int slot, int b) __attribute__((always_inline)) -> int4 {
    const bool from_plane = b == 0 && !p.ll_from_store;
    int v[4];
    if constexpr (DQ8) {
      if (b > 0 && tab_on) { // (wave-uniform) through the table: dequantised at once
        const unsigned w = bq8[slot][b];
        const int *t = tab8 + 256 * (b - 1);
        v[0] = t[w & 0xFFu]; v[1] = t[(w >> 8) & 0xFFu]; v[2] = t[(w >> 16) & 0xFFu]; v[3] = t[w >> 24];
        if (__builtin_expect(min(min(v[0], v[1]), min(v[2], v[3])) == (int)0x80000000, 0)) {
          q_now();
          const int32_t *wq = wide + rec_at(m, b);
#pragma unroll
          for (int k = 0; k < 4; ++k) if (v[k] == (int)0x80000000) v[k] = dequant_full(wq[k], qf[b], qo[b]);
        }
        return make_int4(v[0], v[1], v[2], v[3]);
      }
    }
    if (BP8 && b > 0) {
      const unsigned w = bq8[slot][b];
      v[0] = __builtin_amdgcn_sbfe((int)w, 0, 8); v[1] = __builtin_amdgcn_sbfe((int)w, 8, 8); v[2] = __builtin_amdgcn_sbfe((int)w, 16, 8); v[3] = (int)w >> 24;
      if (min(min(v[0], v[1]), min(v[2], v[3])) == -128) {
        const int32_t *wq = wide + rec_at(m, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (v[k] == -128) v[k] = wq[k];
      }
    } else if constexpr (S_::narrow) {
      const uint2 w = bq[slot][BP8 ? 0 : b];
      v[0] = vc2_lo16(w.x); v[1] = vc2_hi16(w.x); v[2] = vc2_lo16(w.y); v[3] = vc2_hi16(w.y);
      if (min(min(v[0], v[1]), min(v[2], v[3])) == VC2_ST_SENTINEL) {
        const int32_t *wq = from_plane ? llp_w + mul24z(m, ow) : wide + rec_at(m, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (v[k] == VC2_ST_SENTINEL) v[k] = wq[k];
      }
    } else {
      const uint4 w = bq[slot][BP8 ? 0 : b];
      v[0] = (int)w.x; v[1] = (int)w.y; v[2] = (int)w.z; v[3] = (int)w.w;
    }
    if (p.dequant && !from_plane) {
      q_now();
      const int mx = max(max(v[0], v[1]), max(v[2], v[3])), mn = min(min(v[0], v[1]), min(v[2], v[3]));
      if (mx <= ql[b] && mn >= -ql[b]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int sg = min(max(v[k], -1), 1);
          const unsigned t = (__umul24((unsigned)__mul24(v[k], sg), (unsigned)qf[b]) + (unsigned)(qo[b] + 2)) >> 2;
          v[k] = __mul24((int)t, sg);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = dequant_full(v[k], qf[b], qo[b]);
      }
    }
    return make_int4(v[0], v[1], v[2], v[3]);
  };
  // the lane's four coefficients of all four bands of row m (escapes resolved, dequantised): ONE test for escapes and
  // ONE for the dequantiser's fast domain over the sixteen values -- a branch per band cost more than the arithmetic
  auto bands16 = [&](int m, int slot, int (&v)[16]) __attribute__((always_inline)) {
    if constexpr (BP8) { // LL: 16-bit elements of its plane (sentinel -32768); the bands: bytes (sentinel -128)
      const uint2 w0 = bq[slot][0];
      v[0] = vc2_lo16(w0.x); v[1] = vc2_hi16(w0.x); v[2] = vc2_lo16(w0.y); v[3] = vc2_hi16(w0.y);
      if constexpr (DQ8) {
        if (tab_on) { // (wave-uniform) the bands through the table: dequantised at once; LL comes from the level below as it is
#pragma unroll
          for (int b = 1; b < 4; ++b) {
            const unsigned w = bq8[slot][b];
            const int *t = tab8 + 256 * (b - 1);
            v[4 * b] = t[w & 0xFFu]; v[4 * b + 1] = t[(w >> 8) & 0xFFu]; v[4 * b + 2] = t[(w >> 16) & 0xFFu]; v[4 * b + 3] = t[w >> 24];
          }
          int low0 = min(min(v[0], v[1]), min(v[2], v[3])), low = v[4];
#pragma unroll
          for (int k = 5; k < 16; ++k) low = min(low, v[k]);
          if (__builtin_expect(low0 == VC2_ST_SENTINEL || low == (int)0x80000000, 0)) {
            q_now();
            const int32_t *wl = llp_w + mul24z(m, ow);
#pragma unroll
            for (int k = 0; k < 4; ++k) if (v[k] == VC2_ST_SENTINEL) v[k] = wl[k];
#pragma unroll
            for (int b = 1; b < 4; ++b) {
              const int32_t *wq = wide + rec_at(m, b);
#pragma unroll
              for (int k = 0; k < 4; ++k) if (v[4 * b + k] == (int)0x80000000) v[4 * b + k] = dequant_full(wq[k], qf[b], qo[b]);
            }
          }
          return;
        }
      }
#pragma unroll
      for (int b = 1; b < 4; ++b) {
        const unsigned w = bq8[slot][b];
        v[4 * b] = __builtin_amdgcn_sbfe((int)w, 0, 8); v[4 * b + 1] = __builtin_amdgcn_sbfe((int)w, 8, 8);
        v[4 * b + 2] = __builtin_amdgcn_sbfe((int)w, 16, 8); v[4 * b + 3] = (int)w >> 24;
      }
      int low0 = min(min(v[0], v[1]), min(v[2], v[3])), low = v[4];
#pragma unroll
      for (int k = 5; k < 16; ++k) low = min(low, v[k]);
      if (low0 == VC2_ST_SENTINEL || low == -128) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bool from_plane = b == 0 && !p.ll_from_store;
          const int32_t *wq = from_plane ? llp_w + mul24z(m, ow) : wide + rec_at(m, b);
#pragma unroll
          for (int k = 0; k < 4; ++k) if (v[4 * b + k] == (b == 0 ? VC2_ST_SENTINEL : -128)) v[4 * b + k] = wq[k];
        }
      }
    } else if constexpr (S_::narrow) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint2 w = bq[slot][b];
        v[4 * b] = vc2_lo16(w.x); v[4 * b + 1] = vc2_hi16(w.x); v[4 * b + 2] = vc2_lo16(w.y); v[4 * b + 3] = vc2_hi16(w.y);
      }
      int lowest = v[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) lowest = min(lowest, v[k]);
      if (lowest == VC2_ST_SENTINEL) { // (the sentinel is the smallest 16-bit value)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bool from_plane = b == 0 && !p.ll_from_store;
          const int32_t *wq = from_plane ? llp_w + mul24z(m, ow) : wide + rec_at(m, b);
#pragma unroll
          for (int k = 0; k < 4; ++k) if (v[4 * b + k] == VC2_ST_SENTINEL) v[4 * b + k] = wq[k];
        }
      }
    } else if constexpr (!BP8) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint4 w = bq[slot][b];
        v[4 * b] = (int)w.x; v[4 * b + 1] = (int)w.y; v[4 * b + 2] = (int)w.z; v[4 * b + 3] = (int)w.w;
      }
    }
    if (p.dequant) {
      q_now();
      const int bf = p.ll_from_store ? 0 : 1; // bands that come from the store
      bool fast = true;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (b < bf) continue;
        const int mx = max(max(v[4 * b], v[4 * b + 1]), max(v[4 * b + 2], v[4 * b + 3]));
        const int mn = min(min(v[4 * b], v[4 * b + 1]), min(v[4 * b + 2], v[4 * b + 3]));
        fast &= mx <= ql[b] && mn >= -ql[b];
      }
      if (fast) {
        // scale(), Quantisation.cpp:86-95, inside its domain: sign(v) * ((|v| * factor + offset + 2) >> 2), 0 for 0.
        // sg = sign(v) in {-1, 0, 1}; |v| = v * sg; the final product with sg restores the sign and zeroes the v = 0 case
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (b < bf) continue;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int x = v[4 * b + k], sg = min(max(x, -1), 1);
            const unsigned t = (__umul24((unsigned)__mul24(x, sg), (unsigned)qf[b]) + (unsigned)(qo[b] + 2)) >> 2;
            v[4 * b + k] = __mul24((int)t, sg);
          }
        }
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (b < bf) continue;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[4 * b + k] = dequant_full(v[4 * b + k], qf[b], qo[b]);
        }
      }
    }
  };

  // ---- output rows
  const int lim_h = FINAL ? p.pic_h[comp] : out_h;
  uint8_t *rawo = nullptr;
  ST *lvl = nullptr;
  int32_t *lvl_w = nullptr;
  if constexpr (FINAL) rawo = (uint8_t *)p.plane[comp] + (size_t)pic * p.plane_stride[comp] + (size_t)chunk * 16;
  else {
    lvl = (ST *)p.plane[comp] + (size_t)pic * p.plane_stride[comp] + (size_t)chunk * 8;
    if constexpr (S_::narrow) lvl_w = p.plane_wide[comp] + (size_t)pic * p.plane_stride[comp] + (size_t)chunk * 8;
  }
  // an output row: horizontal inverse lifting, rounding, (FINAL) clip / offset / justify / big-endian words.  The row
  // is kept as it will be stored (FINAL, or 16-bit planes: four words; int32 planes: eight) until the next iteration.
  constexpr int OW = (FINAL || S_::narrow) ? 4 : 8;
  struct Pend { unsigned w[OW]; };
  auto make_out = [&](int y, Row &r, Pend &o) __attribute__((always_inline)) {
    h_lift<K, true>(r, redge);
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
        const unsigned a = (unsigned)(min(max(s[2 * k], p.clip_lo), p.clip_hi) + p.sample_offset) << p.sample_shift;
        const unsigned b = (unsigned)(min(max(s[2 * k + 1], p.clip_lo), p.clip_hi) + p.sample_offset) << p.sample_shift;
        o.w[k] = __builtin_amdgcn_perm(b, a, 0x04050001u); // the low halves of a, b as big-endian 16-bit words (pack and swap in one v_perm)
      }
    } else if constexpr (S_::narrow) {
      const int mx = max(max(max(s[0], s[1]), max(s[2], s[3])), max(max(s[4], s[5]), max(s[6], s[7])));
      const int mn = min(min(min(s[0], s[1]), min(s[2], s[3])), min(min(s[4], s[5]), min(s[6], s[7])));
      if ((mx > 32767 || mn < -32767) && own && y < lim_h) { // beyond 16 bits: the wide plane, at once
#pragma unroll
        for (int k = 0; k < 8; ++k) if (!S_::fits(s[k])) { lvl_w[mul24z(y, out_w) + k] = s[k]; s[k] = VC2_ST_SENTINEL; }
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
    if constexpr (FINAL) {
#if VC2_STREAM_NT_OUT
      st_nt(rawo + mul24z(y, out_w) * 2, make_uint4(o.w[0], o.w[1], o.w[2], o.w[3]));
#else
      *(uint4 *)(rawo + mul24z(y, out_w) * 2) = make_uint4(o.w[0], o.w[1], o.w[2], o.w[3]);
#endif
    }
    else if constexpr (S_::narrow) *(uint4 *)(lvl + mul24z(y, out_w)) = make_uint4(o.w[0], o.w[1], o.w[2], o.w[3]);
    else {
      *(uint4 *)(lvl + mul24z(y, out_w)) = make_uint4(o.w[0], o.w[1], o.w[2], o.w[3]);
      *(uint4 *)(lvl + mul24z(y, out_w) + 4) = make_uint4(o.w[OW - 4], o.w[OW - 3], o.w[OW - 2], o.w[OW - 1]);
    }
  };

  // ---- the walk (global memory operations of an iteration in one place, as in the forward kernel)
  VE eng;
  eng.clear();
  const int m0 = max(sp.kA + T::sum_dmin(), 0) & ~(RL - 1);
  const bool last = sp.kB + T::OFFL + RL - 1 > np; // (as in the forward kernel)
  const int mend = last ? np : sp.kB + T::OFFL;
  const int mload = np - 1;
  int sv_have = -1;
#pragma unroll
  for (int u = 0; u < PFI; ++u) in_fetch(min(m0 + u, mload), u);
  static_assert(RL % PFI == 0, "the prefetch ring shares the unrolled walk of the row rings");
  int pend_k = -1;
  Pend pe, po;
#pragma unroll
  for (int k = 0; k < OW; ++k) { pe.w[k] = 0; po.w[k] = 0; }
#define VC2_INV_ITER(U, MODE) VC2_INV_ITERD(U, MODE, U)
#define VC2_INV_ITERD(U, MODE, D)                                                                            \
  if constexpr (MODE != 2 || D < T::OFFL) {                                                                  \
    const int m = MODE == 2 ? np + D : mb + U;                                                               \
    Row re, ro;                                                                                              \
    if constexpr (MODE != 2) {                                                                               \
      const int ml = min(m, mload), sv = ml >> lbsh;                                                         \
      if (sv != sv_have) { load_q(sv); sv_have = sv; }                                                       \
      if constexpr (RL == 4) {                                                                               \
        int bv[16];                                                                                          \
        bands16(ml, U % PFI, bv);                                                                            \
        re = Row{{bv[0], bv[1], bv[2], bv[3], bv[4], bv[5], bv[6], bv[7]}};                                  \
        ro = Row{{bv[8], bv[9], bv[10], bv[11], bv[12], bv[13], bv[14], bv[15]}};                            \
      } else { /* rings of eight: no registers to spare for sixteen values at once */                        \
        const int4 t0 = band4(ml, U % PFI, 0), t1 = band4(ml, U % PFI, 1), t2 = band4(ml, U % PFI, 2), t3 = band4(ml, U % PFI, 3); \
        re = Row{{t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w}};                                          \
        ro = Row{{t2.x, t2.y, t2.z, t2.w, t3.x, t3.y, t3.z, t3.w}};                                          \
      }                                                                                                      \
    }                                                                                                        \
    if (pend_k >= 0) { put_out(2 * pend_k, pe); put_out(2 * pend_k + 1, po); }                               \
    if constexpr (MODE != 2) in_fetch(min(m + PFI, mload), U % PFI);                                         \
    eng.template step<U, MODE, D>(m, np, re, ro);                                                            \
    const int k = m - T::OFFL;                                                                               \
    if (k >= sp.kA && k < sp.kB) {                                                                           \
      Row oe = eng.template out<U>(false), oo = eng.template out<U>(true);                                   \
      make_out(2 * k, oe, pe);                                                                               \
      make_out(2 * k + 1, oo, po);                                                                           \
      pend_k = k;                                                                                            \
    } else pend_k = -1;                                                                                      \
  }
#define VC2_INV_BLOCK(MODE) { VC2_INV_ITER(0, MODE) VC2_INV_ITER(1, MODE) VC2_INV_ITER(2, MODE) VC2_INV_ITER(3, MODE) \
    if constexpr (RL == 8) { VC2_INV_ITER(4, MODE) VC2_INV_ITER(5, MODE) VC2_INV_ITER(6, MODE) VC2_INV_ITER(7, MODE) } }
  int mb = m0;
  if (m0 == 0) {
    VC2_INV_BLOCK(1)
    mb += RL;
  }
  const int mstop = last ? (np & ~(RL - 1)) : mend;
  const int prio0 = VC2_STREAM_WG_WAVES > 1 ? (int)(blockIdx.x >> 8) : (int)(blockIdx.x >> 10); // (the wavefronts that share a SIMD: a chip full of workgroups apart)
  for (; mb < mstop; mb += RL) { if (p.st_prio) prio_turn(prio0 + (mb >> p.st_prio)); VC2_INV_BLOCK(0) }
  if (last) { // as in the forward kernel
#define VC2_INV_DRAIN(R) { VC2_INV_ITERD((R + 0) % RL, 2, 0) VC2_INV_ITERD((R + 1) % RL, 2, 1) VC2_INV_ITERD((R + 2) % RL, 2, 2) \
    VC2_INV_ITERD((R + 3) % RL, 2, 3) VC2_INV_ITERD((R + 4) % RL, 2, 4) VC2_INV_ITERD((R + 5) % RL, 2, 5) VC2_INV_ITERD((R + 6) % RL, 2, 6) }
    if constexpr (RL == 4 && TAIL) {
      switch (np & 3) {
        case 0: VC2_INV_DRAIN(0) break;
        case 1: VC2_INV_ITER(0, 0) VC2_INV_DRAIN(1) break;
        case 2: VC2_INV_ITER(0, 0) VC2_INV_ITER(1, 0) VC2_INV_DRAIN(2) break;
        default: VC2_INV_ITER(0, 0) VC2_INV_ITER(1, 0) VC2_INV_ITER(2, 0) VC2_INV_DRAIN(3) break;
      }
    } else VC2_INV_DRAIN(0)
#undef VC2_INV_DRAIN
  }
#undef VC2_INV_BLOCK
#undef VC2_INV_ITER
#undef VC2_INV_ITERD
  if (pend_k >= 0) { put_out(2 * pend_k, pe); put_out(2 * pend_k + 1, po); }
  VC2_STAMP_END(1);
}

// ------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------
template <int K, bool EDGE, bool INV, class ST, bool TAIL>
void launch_stream(Launcher &L, const LevelParams &p, int n_pictures, size_t lds, hipStream_t s) {
  const int cols = (p.st_strips[0] + p.st_strips[1] + p.st_strips[2]) * n_pictures;
  const int gx = ((cols + 7) / 8) * p.st_segmax * 8; // work items (a multiple of 8)
  constexpr int WW = VC2_STREAM_WG_WAVES;
  dim3 grid(WW > 1 ? ((gx + 8 * WW - 1) / (8 * WW)) * 8 : gx), block(64 * WW);
  LevelParams pw = p;
  pw.st_lds = (int)lds;
  const size_t lds_wg = lds * WW;
#ifdef VC2HIP_STAMPS
  const char *stamp_file = getenv("VC2HIP_STAMPS_FILE");
  const size_t stamp_n = (size_t)gx * 4;
  unsigned long long *d_st = nullptr;
  if (stamp_file) {
    (void)hipMalloc((void **)&d_st, stamp_n * 8);
    (void)hipMemsetAsync(d_st, 0, stamp_n * 8, s);
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_stamps), &d_st, sizeof d_st, 0, hipMemcpyHostToDevice, s);
  }
#endif
  if constexpr (INV) {
    vc2_prof_begin(L, EDGE ? "idwt_level_final" : "idwt_level", s);
    if constexpr (std::is_same<ST, int16_t>::value && !TAIL) {
      if (pw.bp8) {
        vc2_allow_lds((const void *)k_inv_stream<K, EDGE, ST, TAIL, true>, std::max<size_t>(64 * 1024, lds_wg));
        VC2_LAUNCH(L, (k_inv_stream<K, EDGE, ST, TAIL, true>), grid, block, lds_wg, s, pw);
        vc2_prof_end(L, s);
        return;
      }
    }
    vc2_allow_lds((const void *)k_inv_stream<K, EDGE, ST, TAIL>, std::max<size_t>(64 * 1024, lds_wg));
    VC2_LAUNCH(L, (k_inv_stream<K, EDGE, ST, TAIL>), grid, block, lds_wg, s, pw);
  } else {
    vc2_prof_begin(L, EDGE ? "dwt_level_first" : "dwt_level", s);
    vc2_allow_lds((const void *)k_fwd_stream<K, EDGE, ST, TAIL>, std::max<size_t>(64 * 1024, lds_wg));
    VC2_LAUNCH(L, (k_fwd_stream<K, EDGE, ST, TAIL>), grid, block, lds_wg, s, pw);
  }
#ifdef VC2HIP_STAMPS
  if (stamp_file) {
    std::vector<unsigned long long> h(stamp_n);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), d_st, stamp_n * 8, hipMemcpyDeviceToHost);
    unsigned long long *none = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &none, sizeof none);
    (void)hipFree(d_st);
    if (FILE *fp = fopen(stamp_file, "ab")) {
      const int hdr[8] = {INV ? 1 : 0, EDGE ? 1 : 0, gx, 1, 1, (int)lds, 0, 0};
      fwrite(hdr, sizeof hdr, 1, fp);
      fwrite(h.data(), 8, stamp_n, fp);
      fclose(fp);
    }
  }
#endif
  vc2_prof_end(L, s);
}

template <bool INV, class ST> int dispatch_stream(Launcher &L, int kernel, bool edge, const LevelParams &p, int n, size_t lds, hipStream_t s) {
#define VC2_CASE(KK)                                                        \
  case KK:                                                                  \
    if (edge) launch_stream<KK, true, INV, ST, false>(L, p, n, lds, s);     \
    else if (p.st_tail) {                                                   \
      if constexpr (KK != VC2HIP_FIDELITY) launch_stream<KK, false, INV, ST, true>(L, p, n, lds, s); \
    } else launch_stream<KK, false, INV, ST, false>(L, p, n, lds, s);       \
    return 0;
  switch (kernel) {
    VC2_CASE(VC2HIP_DD97)
#ifndef VC2_STREAM_ONE // (quick compiles while working on this file)
    VC2_CASE(VC2HIP_LEGALL)
    VC2_CASE(VC2HIP_DD137)
    VC2_CASE(VC2HIP_HAAR0)
    VC2_CASE(VC2HIP_HAAR1)
    VC2_CASE(VC2HIP_FIDELITY)
    VC2_CASE(VC2HIP_DAUB97)
#endif
  }
#undef VC2_CASE
  return VC2HIP_EINVAL;
}

int halo_lanes_of(int kernel) {
  switch (kernel) {
    case VC2HIP_DD97: return halo_lanes<VC2HIP_DD97>();
    case VC2HIP_LEGALL: return halo_lanes<VC2HIP_LEGALL>();
    case VC2HIP_DD137: return halo_lanes<VC2HIP_DD137>();
    case VC2HIP_HAAR0: return halo_lanes<VC2HIP_HAAR0>();
    case VC2HIP_HAAR1: return halo_lanes<VC2HIP_HAAR1>();
    case VC2HIP_FIDELITY: return halo_lanes<VC2HIP_FIDELITY>();
    case VC2HIP_DAUB97: return halo_lanes<VC2HIP_DAUB97>();
  }
  return -1;
}
bool pow2i(int v) { return v > 0 && (v & (v - 1)) == 0; }

// row pairs a segment walks besides its own: the run-in above (rounded down to a ring block) and the run-out below
template <int K> constexpr int runin_pairs() { return -VT<K, false>::sum_dmin() + VT<K, false>::OFFL + RLK<K>; }
int runin_of(int kernel) {
  switch (kernel) {
    case VC2HIP_DD97: return runin_pairs<VC2HIP_DD97>();
    case VC2HIP_LEGALL: return runin_pairs<VC2HIP_LEGALL>();
    case VC2HIP_DD137: return runin_pairs<VC2HIP_DD137>();
    case VC2HIP_HAAR0: return runin_pairs<VC2HIP_HAAR0>();
    case VC2HIP_HAAR1: return runin_pairs<VC2HIP_HAAR1>();
    case VC2HIP_FIDELITY: return runin_pairs<VC2HIP_FIDELITY>();
    case VC2HIP_DAUB97: return runin_pairs<VC2HIP_DAUB97>();
  }
  return 8;
}
// wavefront slots of the device for one streaming kernel: the runtime's answer per CU, in whole wavefronts per SIMD
// (one-wavefront workgroups are dealt to the four SIMDs in turn: 13 per CU by LDS ran as 12), times the CUs
template <int K, bool EDGE, bool INV, class ST, bool TAIL> int slots_of(size_t lds) {
  int nb = 0, dev = 0;
  hipDeviceProp_t prop;
  const void *fn = INV ? (const void *)k_inv_stream<K, EDGE, ST, TAIL> : (const void *)k_fwd_stream<K, EDGE, ST, TAIL>;
  constexpr int WW = VC2_STREAM_WG_WAVES;
  vc2_allow_lds(fn, std::max<size_t>(64 * 1024, lds * WW));
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * WW, lds * WW) != hipSuccess || nb < 1)
    return 256 * 8;
  if (WW == 1 && nb >= 4) nb &= ~3;
  return nb * WW * prop.multiProcessorCount; // (workgroups of four wavefronts: one per SIMD)
}
template <bool INV, class ST> int slots_dispatch(int kernel, bool edge, bool tail, size_t lds) {
#define VC2_CASE(KK)                                                                       \
  case KK:                                                                                 \
    if (edge) return slots_of<KK, true, INV, ST, false>(lds);                              \
    if (tail) { if constexpr (KK != VC2HIP_FIDELITY) return slots_of<KK, false, INV, ST, true>(lds); return 2048; } \
    return slots_of<KK, false, INV, ST, false>(lds);
  switch (kernel) {
    VC2_CASE(VC2HIP_DD97)
#ifndef VC2_STREAM_ONE // (quick compiles while working on this file)
    VC2_CASE(VC2HIP_LEGALL)
    VC2_CASE(VC2HIP_DD137)
    VC2_CASE(VC2HIP_HAAR0)
    VC2_CASE(VC2HIP_HAAR1)
    VC2_CASE(VC2HIP_FIDELITY)
    VC2_CASE(VC2HIP_DAUB97)
#endif
  }
#undef VC2_CASE
  return 2048;
}
int stream_slots(int kernel, bool edge, bool inverse, bool store16, bool tail, size_t lds) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int, int, size_t, int>, int> cache; // (per device: one process may drive several GPUs)
  int dev = 0;
  (void)hipGetDevice(&dev);
  const auto key = std::make_tuple(kernel, (int)edge, (int)inverse, (int)store16, (int)tail, lds, dev);
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int v;
  if (inverse) v = store16 ? slots_dispatch<true, int16_t>(kernel, edge, tail, lds) : slots_dispatch<true, int32_t>(kernel, edge, tail, lds);
  else v = store16 ? slots_dispatch<false, int16_t>(kernel, edge, tail, lds) : slots_dispatch<false, int32_t>(kernel, edge, tail, lds);
  cache[key] = v;
  return v;
}

} // namespace

// The streaming kernels apply when every active component has a plane at least 64 chunks wide without horizontal
// padding, power-of-two slice footprints of at least one chunk, band blocks that move in whole 16-byte pieces, and
// raw samples (edge levels) in 16-bit words.  Fills the st_* fields of p and returns the dynamic LDS bytes, 0 if not
// applicable.  (Fidelity's 8-tap steps keep ~21 rows of 8 values per lane alive: rings of 8, two wavefronts per SIMD.)
size_t vc2_stream_level_applicable(LevelParams &p, int kernel, bool edge, bool inverse, bool store16, int n_pictures) {
  const int hln = halo_lanes_of(kernel);
  if (hln < 0) return 0;
  const int elem = store16 ? 2 : 4, ep = 16 / elem;
  size_t lds = 0;
  p.st_tail = 0;
  p.st_segmax = 0;
  for (int c = 0; c < 3; ++c) {
    p.st_strips[c] = p.st_segs[c] = 0;
    if (p.tiles_x[c] == 0 || p.tiles_y[c] == 0) continue;
    const int w = p.in_w[c], h = p.in_h[c], fw = p.fw[c], fh = p.fh[c];
    const int rl = kernel == VC2HIP_FIDELITY ? 8 : 4;
    // whole blocks of `rl` row pairs; below the first / last level, rings of four: any even height (the TAIL kernels)
    if (w < VC2_STREAM_MIN_W || (w & 7) || (h & 1) || h < 4 * rl) return 0;
    if (h % (2 * rl)) {
      if (edge || rl == 8) return 0;
      p.st_tail = 1;
    }
    if (edge && (p.word_bytes != 2 || p.pic_w[c] != w)) return 0;
    if (!pow2i(fw) || !pow2i(fh) || fw < 8 || fh < 2 || fw > 64 * 8) return 0;
    const int bsh = fh / 2, bsw = fw / 2;
    if (bsh * bsw < ep || (bsw < ep && bsw * 2 != ep)) return 0;
    if ((p.coef_off[c] % ep) || (p.slice_coefs % ep) || ((bsh * bsw) % ep)) return 0;
    const int lps = fw / 8;
    const int out = ((64 - 2 * hln) / lps) * lps;
    if (out < lps) return 0;
    const int nch = w / 8;
    p.st_out[c] = out;
    p.st_llps[c] = 31 - __builtin_clz((unsigned)lps);
    p.st_strips[c] = (nch + out - 1) / out;
    // image: forward out * 4 elements per row; inverse every slice the 64 chunks touch
    const size_t rs = (size_t)out * 4;
    size_t img;
    if (!inverse) img = (p.ll_to_store ? 4 : 3) * (size_t)bsh * rs * elem;
    else img = 16; // the inverse kernel reads the records directly
#ifdef VC2_STREAM_DIRECT
    img = 16;
#endif
    lds = std::max(lds, img);
  }
  { // (ablation build: extra LDS per wavefront -- what the kernels lose with fewer resident wavefronts)
    static const int pad4[4] = {vc2_tune_int("VC2HIP_STREAM_LDSPAD_FF", 0), vc2_tune_int("VC2HIP_STREAM_LDSPAD_FL", 0),
                                vc2_tune_int("VC2HIP_STREAM_LDSPAD_IF", 0), vc2_tune_int("VC2HIP_STREAM_LDSPAD_IL", 0)};
    lds += (size_t)pad4[(inverse ? 2 : 0) + (edge ? 0 : 1)];
  }
  if (lds > 40 * 1024 || lds * VC2_STREAM_WG_WAVES > 144 * 1024) return 0;
  // Segments: whole rows of slices, `nseg` per strip, the same for every component; segment g of a strip covers the
  // slice rows [g * ys / nseg, (g + 1) * ys / nseg) -- heights differ by at most one slice row.  A segment runs in over
  // the filter's reach before its first output row (about 8 row pairs for DD97, 22 for Fidelity): short segments waste
  // work, long ones leave the chip without wavefronts, and a launch whose wavefronts do not fill whole rounds of the
  // chip's wavefront slots idles through its last round (round 2: 8704 equal wavefronts for 4096 slots).  So: the nseg
  // that minimises rounds * (rows of the tallest segment + run-in), rounds = ceil(strips * nseg / slots).
  static const int force_nseg = vc2_tune_int("VC2HIP_STREAM_NSEG", 0);
  int cols = 0, bsh_max = 1;
  for (int c = 0; c < 3; ++c) if (p.st_strips[c]) { cols += p.st_strips[c] * n_pictures; bsh_max = std::max(bsh_max, p.fh[c] / 2); }
  const int slots = stream_slots(kernel, edge, inverse, store16, p.st_tail != 0, lds);
  const int runin = runin_of(kernel);
  int nseg = 1;
  long long best = -1;
  for (int g = 1; g <= p.ys; ++g) {
    const long long rounds = ((long long)cols * g + slots - 1) / slots;
    const long long cost = rounds * ((long long)((p.ys + g - 1) / g) * bsh_max + runin);
    if (best < 0 || cost <= best) { best = cost; nseg = g; } // (ties: more wavefronts)
  }
  if (force_nseg > 0) nseg = std::min(force_nseg, p.ys);
  { // (ablation build: one of the four kernels only -- forward first / forward level / inverse final / inverse level)
    static const int f4[4] = {vc2_tune_int("VC2HIP_STREAM_NSEG_FF", 0), vc2_tune_int("VC2HIP_STREAM_NSEG_FL", 0),
                              vc2_tune_int("VC2HIP_STREAM_NSEG_IF", 0), vc2_tune_int("VC2HIP_STREAM_NSEG_IL", 0)};
    const int f = f4[(inverse ? 2 : 0) + (edge ? 0 : 1)];
    if (f > 0) nseg = std::min(f, p.ys);
  }
#ifdef VC2HIP_STAMPS
  fprintf(stderr, "stream level: kernel %d edge %d inv %d lds %zu cols %d slots %d runin %d -> nseg %d\n", kernel, (int)edge, (int)inverse, lds, cols, slots, runin, nseg);
#endif
  for (int c = 0; c < 3; ++c) if (p.st_strips[c]) p.st_segs[c] = nseg;
  p.st_segmax = nseg;
  p.st_npic = n_pictures;
  static const int prio = vc2_tune_int("VC2HIP_STREAM_PRIO", 2);
  p.st_prio = prio; // a new turn at the highest priority every four row pairs (measured: 2, 3 and 4 alike; off: inverse level 0 9 % slower)
  return lds;
}
int vc2_launch_forward_stream(Launcher &L, int kernel, bool first, const LevelParams &p, int n, bool store16, size_t lds, hipStream_t s) {
  return store16 ? dispatch_stream<false, int16_t>(L, kernel, first, p, n, lds, s) : dispatch_stream<false, int32_t>(L, kernel, first, p, n, lds, s);
}
int vc2_launch_inverse_stream(Launcher &L, int kernel, bool final_level, const LevelParams &p, int n, bool store16, size_t lds, hipStream_t s) {
  return store16 ? dispatch_stream<true, int16_t>(L, kernel, final_level, p, n, lds, s) : dispatch_stream<true, int32_t>(L, kernel, final_level, p, n, lds, s);
}
