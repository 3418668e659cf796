// Quantisation + HQ / LD slice coding kernels.
//
// Replaces, from /root/reference/src/Library/src:
//   quant / quant_factor / adjust_quant_index        Quantisation.cpp:16-20, :40-76
//   component_slice_bytes                            Slices.cpp:97-119
//   HQSliceIO_VBR / HQSliceIO_CBR (out and in)       Slices.cpp:305-612 (over :645-694)
//   LDSliceIO (in)                                   Slices.cpp:246-303
//   interleaved exp-Golomb, bounded bit I/O          VLC.cpp:21-94, :151-257
//   quantIndicesCBR + yss_for_slice                  EncodeStream.cpp:73-125, Quantisation.cpp:627-642
//   inverse_quantise_LLSubband / predictDC           Quantisation.cpp:191-208, :287-306
//
// Work unit = one slice (HQ pack, CBR search: one wavefront per slice; unpack: one lane per
// slice component).  The coefficient store keeps each slice's coefficients contiguous and in
// coding order, so reads and writes here are plain streams.
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>

#include "vc2hip_internal.h"
#include "vc2hip_store.h"

void vc2_prof_begin(Launcher &L, const char *name, hipStream_t s);
void vc2_prof_end(Launcher &L, hipStream_t s);

__constant__ QuantTables c_qs;
void vc2_upload_p16_tables(const QuantTables &t, hipStream_t s);
void vc2_upload_tables_slices(const QuantTables &t, hipStream_t s) {
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(c_qs), &t, sizeof t, 0, hipMemcpyHostToDevice, s);
  vc2_upload_p16_tables(t, s);
}

// ------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------
// Quantisation.cpp:69-76
// |v| << 2 divided by qf: exact reciprocal multiply inside the reference's domain (qf > 0, no int
// overflow); outside it the literal int division.  The slow branch is taken wave-uniformly so that
// the 25-instruction division sequence is not if-converted into the common path.
__device__ __forceinline__ int quant_core(int v, int qf, unsigned mg, int sh) {
  const unsigned mag = v < 0 ? 0u - (unsigned)v : (unsigned)v;
  const int a = (int)(mag << 2);
  const unsigned t = __umulhi(mg, (unsigned)a);
  int q = (int)((t + (((unsigned)a - t) >> 1)) >> sh);
  const bool slow = !(qf > 1 && a >= 0);
  if (__any(slow)) { if (slow) q = a / qf; }
  return v < 0 ? (int)(0u - (unsigned)q) : q;
}
__device__ __forceinline__ int quant_dev(int v, int aq) {
  return quant_core(v, c_qs.qf[aq], c_qs.magic[aq], c_qs.shift[aq]);
}
// band index of coefficient j when n0 is (usually) a power of two: no integer division
__host__ __device__ __forceinline__ int band_of_index_fast(int j, int n0, int n0_shift) {
  const int m = n0_shift >= 0 ? (j >> n0_shift) : (j / n0);
  if (m == 0) return 0;
  const int L = (31 - __builtin_clz((unsigned)m)) / 2 + 1;
  return 3 * (L - 1) + (m >> (2 * (L - 1)));
}
// subband of every coefficient index of the common geometry (components of up to 512 / 256 coefficients): 512 luma
// entries, 256 chroma entries.  Filled by the launchers, carried in the kernel arguments, copied into LDS by every
// workgroup (computing it there cost every workgroup of four slices ~45 instructions per thread).
static void fill_band_lut(unsigned char *lut, const int comp_n[3], const int comp_n0[3]) {
  const int n0y = comp_n0[0], n0c = comp_n0[1];
  const int sy = n0y > 0 && (n0y & (n0y - 1)) == 0 ? 31 - __builtin_clz((unsigned)n0y) : -1;
  const int sc = n0c > 0 && (n0c & (n0c - 1)) == 0 ? 31 - __builtin_clz((unsigned)n0c) : -1;
  for (int j = 0; j < 512; ++j) lut[j] = n0y > 0 ? (unsigned char)band_of_index_fast(std::min(j, comp_n[0] - 1), n0y, sy) : 0;
  for (int j = 0; j < 256; ++j) lut[512 + j] = n0c > 0 ? (unsigned char)band_of_index_fast(std::min(j, comp_n[1] - 1), n0c, sc) : 0;
}
__device__ __forceinline__ void copy_band_lut(unsigned char *band_y, const unsigned char *arg) { // band_c = band_y + 512
  for (int j = threadIdx.x; j < 192; j += blockDim.x) ((unsigned *)band_y)[j] = ((const unsigned *)arg)[j];
}
// Quantisation.cpp:86-95
__device__ __forceinline__ int scale_dev(int v, int aq) {
  const int qf = c_qs.qf[aq], off = c_qs.off[aq];
  const unsigned mag = v < 0 ? 0u - (unsigned)v : (unsigned)v;
  int a = (int)(mag * (unsigned)qf);
  if (a > 0) a = (int)((unsigned)a + (unsigned)off);
  a = (int)((unsigned)a + 2u);
  a /= 4;
  return v < 0 ? (int)(0u - (unsigned)a) : a;
}

// SignedVLC(v).numOfBits(), VLC.cpp:78-85: 1 for 0, else 2*floor(log2(|v|+1)) + 2
__device__ __forceinline__ int svlc_bits(int v) {
  if (v == 0) return 1;
  const unsigned m = (v < 0 ? 0u - (unsigned)v : (unsigned)v) + 1u;
  return 2 * (31 - __clz(m)) + 2;
}
// spread the low 16 bits of x to the even bit positions
__device__ __forceinline__ unsigned spread16(unsigned x) {
  x &= 0xFFFFu;
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}
// SignedVLC(v).code(), VLC.cpp:21-52, :78-85 (valid for |v| <= 65534, the reference's own domain)
__device__ __forceinline__ unsigned svlc_code(int v) {
  if (v == 0) return 1u;
  const unsigned m = (v < 0 ? 0u - (unsigned)v : (unsigned)v) + 1u;
  const int k = 31 - __clz(m);
  const unsigned low = m & ((1u << k) - 1u);
  const unsigned u = (spread16(low) << 1) | 1u; // (0 b)* 1
  return (u << 1) | (v < 0 ? 1u : 0u);
}

// Scans and reductions over the lanes of a wavefront (or of its segments of W consecutive lanes) with DPP adds:
// row_shr 1, 2, 4, 8 inside each row of 16 lanes, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into
// rows 2 and 3 -- one VALU instruction per step instead of a ds_bpermute round trip.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp0(int v) { // lanes without a source read 0
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
// inclusive sum over segments of W = 16, 32 or 64 lanes
template <int W> __device__ __forceinline__ int seg_incl_scan(int v, int /*sl*/) {
  static_assert(W == 16 || W == 32 || W == 64, "segments are whole DPP rows");
  v += dpp0<0x111, 0xf>(v);
  v += dpp0<0x112, 0xf>(v);
  v += dpp0<0x114, 0xf>(v);
  v += dpp0<0x118, 0xf>(v);
  if constexpr (W >= 32) v += dpp0<0x142, 0xa>(v);
  if constexpr (W >= 64) v += dpp0<0x143, 0xc>(v);
  return v;
}
// maximum of NON-NEGATIVE values over each segment of W lanes, in every lane of the segment
template <int W> __device__ __forceinline__ int seg_max(int v) {
  if constexpr (W < 16) { // half of a 16-lane segment: butterfly
#pragma unroll
    for (int d = W / 2; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, W));
    return v;
  } else {
    v = max(v, dpp0<0x111, 0xf>(v));
    v = max(v, dpp0<0x112, 0xf>(v));
    v = max(v, dpp0<0x114, 0xf>(v));
    v = max(v, dpp0<0x118, 0xf>(v));
    if constexpr (W >= 32) v = max(v, dpp0<0x142, 0xa>(v));
    if constexpr (W >= 64) v = max(v, dpp0<0x143, 0xc>(v));
    // the last lane of the segment holds the maximum
    if constexpr (W == 64) return __builtin_amdgcn_readlane(v, 63);
    else return __shfl(v, (int)(__lane_id() | (W - 1)));
  }
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) { return seg_incl_scan<64>(v, lane); }
__device__ __forceinline__ int wave_max(int v) { return seg_max<64>(v); }
__device__ __forceinline__ long long wave_sum64(long long v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// LDS written by some lanes of a wavefront and read by others of the SAME wavefront: the hardware executes a
// wavefront's LDS operations in order, this only keeps the compiler from moving accesses across the hand-over
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// OR a code of nb bits at MSB-first bit position pos of a big-endian-word bit buffer
__device__ __forceinline__ void put_code(unsigned *buf, int pos, unsigned code, int nb) {
  const int wi = pos >> 5, bo = pos & 31;
  const unsigned long long v = (unsigned long long)code << (64 - nb - bo);
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  if (hi) atomicOr(&buf[wi], hi);
  if (lo) atomicOr(&buf[wi + 1], lo);
}

// One component of one slice, processed by one wavefront: (optionally) quantise, measure the
// code lengths, (optionally) write the codes into `bits`, and return the number of bits up to
// and including the last non-zero coefficient (component_slice_bytes' `count`).
template <bool QUANT, bool WRITE, class Src>
__device__ __forceinline__ int component_bits(Src src, int n, int n0, int q, const int *qm, int lane,
                                              unsigned *bits, int region_bits, unsigned *err, int bit0 = 0) {
  int base = 0, count = 0;
  const int n0_shift = (n0 & (n0 - 1)) == 0 ? 31 - __clz(n0) : -1;
  for (int r0 = 0; r0 < n; r0 += 512) {
    int v[8], nb[8];
    int sum = 0, last_end = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int j = r0 + lane * 8 + k;
      int c = 0;
      if (j < n) {
        c = src(j);
        if (QUANT) {
          const int aq = max(q - qm[band_of_index_fast(j, n0, n0_shift)], 0);
          if (aq > 119) { atomicOr(err, VC2_DEVERR_QINDEX); c = 0; }
          else c = quant_dev(c, aq);
        }
        nb[k] = svlc_bits(c);
        // a code of more than 32 bits (|c| > 65534, VLC.h:27) cannot be written; MEASURED it has its length, as
        // SignedVLC::numOfBits has in luma_slice_bits (Slices.cpp:51-70): the trials of a quantiser search that meet one
        // simply do not fit (found by tools/fuzz_geometry.py: 16-bit noise, 2 x 2 slices, HQ_CBR)
        if (WRITE && nb[k] > 32) { atomicOr(err, VC2_DEVERR_CODE32); c = 0; nb[k] = 1; }
      } else {
        nb[k] = 0;
      }
      v[k] = c;
      sum += nb[k];
      if (c != 0) last_end = sum;
    }
    const int incl = wave_incl_scan(sum, lane);
    const int lane_base = base + incl - sum;
    count = max(count, wave_max(last_end ? lane_base + last_end : 0));
    if (WRITE) {
      int pos = lane_base;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (nb[k] && pos + nb[k] <= region_bits) put_code(bits, bit0 + pos, svlc_code(v[k]), nb[k]);
        pos += nb[k];
      }
    }
    base += __shfl(incl, 63);
  }
  return count;
}

// component_bits<true, false> for coefficients staged in LDS: a lane's eight consecutive coefficients are read
// as two 16-byte words (eight scalar reads at a 32-byte lane stride would hit every bank eight times over)
__device__ __forceinline__ int component_bits_lds(const int *src, int n, int n0, int q, const int *qm, int lane, unsigned *err) {
  int base = 0, count = 0;
  const int n0_shift = (n0 & (n0 - 1)) == 0 ? 31 - __clz(n0) : -1;
  const bool aligned = (((size_t)src) & 15) == 0;
  for (int r0 = 0; r0 < n; r0 += 512) {
    const int j0 = r0 + lane * 8;
    int v[8];
    if (aligned && j0 + 8 <= n) {
      const int4 a = *(const int4 *)(src + j0), b = *(const int4 *)(src + j0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = j0 + k < n ? src[j0 + k] : 0;
    }
    int sum = 0, last_end = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int j = j0 + k;
      int nb = 0;
      if (j < n) {
        const int aq = max(q - qm[band_of_index_fast(j, n0, n0_shift)], 0);
        int c = 0;
        if (aq > 119) atomicOr(err, VC2_DEVERR_QINDEX);
        else c = quant_dev(v[k], aq);
        nb = svlc_bits(c); // (measured only: more than 32 bits is a length like any other, see component_bits)
        sum += nb;
        if (c != 0) last_end = sum;
      }
    }
    const int incl = wave_incl_scan(sum, lane);
    count = max(count, wave_max(last_end ? base + incl - sum + last_end : 0));
    base += __shfl(incl, 63);
  }
  return count;
}

// ------------------------------------------------------------------------------------------
// HQ pack: one wavefront per slice.  The slice's byte image (prefix, qindex, then per component
// a length byte + bounded exp-Golomb data, Slices.cpp:469-533 / :305-382) is assembled in LDS as
// big-endian words and copied out with dword stores.  Each lane owns 8 consecutive coefficients,
// merges their codes in registers and ORs whole words into the image.
// ------------------------------------------------------------------------------------------
struct Coef8 {
  unsigned code[8]; // exp-Golomb code of each coefficient (right aligned)
  int nb[8];        // its length in bits (0: past the end of the component)
  int sum, last_end;
};
// (code << 6 | length) of the signed exp-Golomb code of +m for m < VLC_LUT_N: computed once on the
// host, kept in device memory, copied into LDS by every pack workgroup (4 KiB).  Larger magnitudes
// (up to the reference's 65534 limit) take the arithmetic path.
constexpr int VLC_LUT_N = 1024; // (A/B on the bench pictures: 4096 entries 0.433 ms, 2048 0.413, 1024 0.408, 512 0.449 -- the copy into LDS against the arithmetic path)
__device__ unsigned g_vlc_lut[VLC_LUT_N];
__device__ __forceinline__ void build_vlc_lut(unsigned *lut) {
  for (int m = threadIdx.x * 4; m < VLC_LUT_N; m += blockDim.x * 4) *(uint4 *)(lut + m) = *(const uint4 *)(g_vlc_lut + m);
}
static void vc2_upload_vlc_lut_s(hipStream_t s);
void vc2_upload_vlc_lut(hipStream_t s) {
  static unsigned host[VLC_LUT_N]; // built once, see vc2_upload_unpack_lut
  static std::once_flag once;
  std::call_once(once, [] {
  for (unsigned m = 0; m < (unsigned)VLC_LUT_N; ++m) {
    unsigned code = 1, nb = 1;
    if (m) {
      const unsigned v = m + 1;
      int k = 31;
      while (!((v >> k) & 1u)) --k;
      code = 0;
      for (int b = k - 1; b >= 0; --b) code = (code << 2) | ((v >> b) & 1u);
      code = ((code << 1) | 1u) << 1; // terminator, then sign bit 0
      nb = 2 * (unsigned)k + 2;
    }
    host[m] = (code << 6) | nb;
  }
  });
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_vlc_lut), host, sizeof host, 0, hipMemcpyHostToDevice, s);
  vc2_upload_vlc_lut_s(s);
}
__device__ __forceinline__ void codes8(Coef8 &c, int (&raw)[8], int j0, int n, unsigned *err, const unsigned *lut);

// load (and quantise) the 8 coefficients [j0, j0+8) of a component record
template <bool QUANT, class ST>
__device__ __forceinline__ void load8(Coef8 &c, const ST *src, const int32_t *wide, int j0, int n, int n0, int n0_shift,
                                      const uint4 *qtab, unsigned *err, const unsigned *lut) {
  c.sum = 0;
  c.last_end = 0;
  int raw[8];
  const bool full = j0 + 8 <= n;
  if (full) St<ST>::load8(src + j0, wide + j0, raw);
  else {
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = j0 + k < n ? St<ST>::load1(src + j0 + k, wide + j0 + k) : 0;
  }
  if (QUANT) { // quantiser constants of the slice's index per subband: qtab (LDS), see k_hq_pack
    const int b0 = band_of_index_fast(min(j0, n - 1), n0, n0_shift), b7 = band_of_index_fast(min(j0 + 7, n - 1), n0, n0_shift);
    if (b0 == b7) { // the usual case: all eight coefficients in one subband
      const uint4 t = qtab[b0];
#pragma unroll
      for (int k = 0; k < 8; ++k) raw[k] = quant_core(raw[k], (int)t.z, t.x, (int)t.y);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (j0 + k >= n) continue;
        const uint4 t = qtab[band_of_index_fast(j0 + k, n0, n0_shift)];
        raw[k] = quant_core(raw[k], (int)t.z, t.x, (int)t.y);
      }
    }
  }
  codes8(c, raw, j0, n, err, lut);
}

// exp-Golomb codes + lengths of eight quantised values (table for |v| < 256, computed beyond)
__device__ __forceinline__ void codes8(Coef8 &c, int (&raw)[8], int j0, int n, unsigned *err, const unsigned *lut) {
  bool big = false;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int v = raw[k];
    const unsigned mag = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    const unsigned e = lut[mag & (VLC_LUT_N - 1)]; // positive code; a negative value sets the sign (last) bit
    c.code[k] = (e >> 6) | (v < 0 ? 1u : 0u);
    c.nb[k] = j0 + k < n ? (int)(e & 63u) : 0;
    big |= mag >= (unsigned)VLC_LUT_N;
  }
  if (__any(big)) { // rare: coefficients beyond the table
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int v = raw[k];
      const unsigned mag = v < 0 ? 0u - (unsigned)v : (unsigned)v;
      if (mag >= (unsigned)VLC_LUT_N && j0 + k < n) {
        int nb = svlc_bits(v);
        unsigned code = svlc_code(v);
        if (nb > 32) { atomicOr(err, VC2_DEVERR_CODE32); nb = 1; code = 1; raw[k] = 0; }
        c.nb[k] = nb;
        c.code[k] = code;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    c.sum += c.nb[k];
    if (raw[k] != 0 && c.nb[k]) c.last_end = c.sum;
  }
}

// Table-driven variant for the common geometry (component <= 512 coefficients): the subband of every
// coefficient index comes from a per-workgroup byte table, the quantiser constants of every subband
// from a per-slice table, both in LDS -- no per-coefficient band arithmetic, no divergence between
// lanes whose eight coefficients straddle subbands and lanes whose do not.
// PRE: the eight 16-bit elements were fetched earlier (`pre`, valid when j0 + 8 <= n) -- the packer issues the luma round's
// loads before its table set-up, so that their latency and the tables' run side by side.
template <class ST, bool PRE = false>
__device__ __forceinline__ void load8_tab(Coef8 &c, const ST *src, const int32_t *wide, int j0, int n, const unsigned char *band_lut,
                                          const uint4 *qtab, unsigned *err, const unsigned *lut, const uint4 pre = make_uint4(0u, 0u, 0u, 0u)) {
  c.sum = 0;
  c.last_end = 0;
  int raw[8];
  if (j0 + 8 <= n) {
    if constexpr (PRE && St<ST>::narrow) St<int16_t>::unpack8(pre, wide + j0, raw);
    else St<ST>::load8(src + j0, wide + j0, raw);
    const uint2 bands = *(const uint2 *)(band_lut + j0);
    // quant(), Quantisation.cpp:69-76, eight at a time: floor(4|v| / factor) as the truncated float product of |v| and
    // the rounded-up 4 / factor (exact for |v| < 2^20, see k_cbr_search_reg), then ONE test whether any left that domain
    int qq[8];
    float big = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned b = ((k < 4 ? bands.x : bands.y) >> (8 * (k & 3))) & 0xFFu;
      const float f = (float)raw[k];
      qq[k] = (int)(unsigned)(__builtin_fabsf(f) * __uint_as_float(qtab[b].w));
      big = fmaxf(big, __builtin_fabsf(f));
    }
    if (__any(big >= 1048576.f)) { // rare: the literal int division for the coefficients concerned
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned b = ((k < 4 ? bands.x : bands.y) >> (8 * (k & 3))) & 0xFFu;
        const unsigned a = (raw[k] < 0 ? 0u - (unsigned)raw[k] : (unsigned)raw[k]) << 2;
        if (a >= (1u << 22)) qq[k] = (int)a / (int)qtab[b].z;
      }
    }
    // codes straight from magnitude and sign (the table holds the code of +m; a negative value sets its last bit)
    unsigned all = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned m = (unsigned)qq[k], e = lut[m & (VLC_LUT_N - 1)];
      c.code[k] = (e >> 6) | ((unsigned)raw[k] >> 31);
      c.nb[k] = (int)(e & 63u);
      all |= m;
    }
    if (!__any(all >= (unsigned)VLC_LUT_N)) { // every magnitude inside the table (and no negative quotient)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        c.sum += c.nb[k];
        if (qq[k] != 0) c.last_end = c.sum;
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = raw[k] < 0 ? (int)(0u - (unsigned)qq[k]) : qq[k];
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      raw[k] = 0;
      if (j0 + k < n) { const uint4 t = qtab[band_lut[j0 + k]]; raw[k] = quant_core(St<ST>::load1(src + j0 + k, wide + j0 + k), (int)t.z, t.x, (int)t.y); }
    }
  }
  codes8(c, raw, j0, n, err, lut);
}

// merge the lane's codes and OR them into the image at absolute bit position pos (codes that
// would cross `limit` are dropped: by construction they are trailing '1's, VLC.cpp:151-156)
__device__ __forceinline__ void write8(unsigned *img, int pos, int limit, const Coef8 &c, bool skip = false) {
  if (skip) return;
  int wi = pos >> 5, fill = pos & 31;
  unsigned long long acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int nb = c.nb[k];
    if (nb != 0 && pos + nb <= limit) { // once one code is past the limit, every later one is too
      acc |= (unsigned long long)c.code[k] << (64 - fill - nb);
      fill += nb;
      if (fill >= 32) {
        atomicOr(&img[wi], (unsigned)(acc >> 32));
        acc <<= 32;
        fill -= 32;
        ++wi;
      }
    }
    pos += nb;
  }
  if (acc) atomicOr(&img[wi], (unsigned)(acc >> 32));
}
// The same for a wavefront none of whose lanes has more than 64 bits of codes (what quantised pictures look like: 36 bits per
// eight coefficients at the bench's index): the lane's eight codes become ONE string in a register pair -- a shift and an
// OR per code, no test per code whether a word is full -- which is cut at the limit (what lies beyond it are the '1's of
// trailing zeros) and OR-ed into the three words it can touch.  write8 above remains for every other wavefront.
__device__ __forceinline__ void write8_short(unsigned *img, int pos, int limit, const Coef8 &c) {
  unsigned long long acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc = (acc << c.nb[k]) | (c.nb[k] ? c.code[k] : 0u);
  int n = c.sum;
  if (pos + n > limit) { const int keep = max(limit - pos, 0); acc >>= (n - keep); n = keep; }
  if (n <= 0) return;
  const int wi = pos >> 5, bo = pos & 31;
  const unsigned long long v = acc << (64 - n); // left aligned
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned w0 = hi >> bo;
  const unsigned w1 = bo ? (hi << (32 - bo)) | (lo >> bo) : lo;
  const unsigned w2 = bo ? lo << (32 - bo) : 0u;
  if (w0) atomicOr(&img[wi], w0);
  if (w1) atomicOr(&img[wi + 1], w1);
  if (w2) atomicOr(&img[wi + 2], w2);
}
__device__ __forceinline__ void put_byte(unsigned *img, int off, unsigned b) { atomicOr(&img[off >> 2], b << (24 - 8 * (off & 3))); }


template <class ST, bool TRIAL = false>
__device__ __forceinline__ void bits8_tab(const ST *src, const int32_t *wide, int j0, int n, const unsigned char *band_lut,
                                          const uint4 *qtab, unsigned *err, int &sum, int &last_end);

// W lanes work on one slice: 64 (a wavefront per slice, any geometry) or, for small slices, 32 / 16 with two / four
// slices per wavefront, so that a slice of e.g. 128 + 2 x 64 coefficients (1080p, -u 2 -a 4) still fills its lanes.
// GIMG: the slice images live in the slots in global memory instead of LDS (slices too long for LDS: slice size scalars
// beyond ~160; VBR and CBR then both go through slots + compaction) -- the same code on flat atomics, an order slower, for
// command lines the reference accepts and nobody uses.
#ifdef VC2HIP_STAMPS
#include <stdio.h>
#include <vector>
__device__ unsigned long long *g_pack_stamps;
#ifndef VC2_STAMP_CLOCK
#define VC2_STAMP_CLOCK wall_clock64 // 100 MHz; -DVC2_STAMP_CLOCK=clock64 stamps shader-clock cycles instead (2.27 GHz under this kernel)
#endif
#define PACK_STAMP(k) do { if (threadIdx.x == 0 && g_pack_stamps && blockIdx.y == 0) g_pack_stamps[8 * (size_t)blockIdx.x + (k)] = VC2_STAMP_CLOCK(); } while (0)
#else
#define PACK_STAMP(k)
#endif
template <int W, bool MID, class ST, bool GIMG = false>
__global__ __launch_bounds__(256) void k_hq_pack(const PackParams p) {
  constexpr int S = 64 / W;
  // (the ablation build's switches.  "No code writes" moves the components' limits to bit 0 instead of branching around the
  // writers: with a branch there -- never taken, the switch being off -- the ablation build of round 4 and 5 lost the leading
  // luma codes of the second to fourth slice of a four-slice wavefront in ~6 % of 2 x 2-sample slices, different ones in every
  // run (tools/probe/small_slices_diff.py; bisected to exactly that branch); the release build never had it)
  const bool skip_codes = VC2_SKIP(p, 1), skip_out = VC2_SKIP(p, 2);
  const int nwv = (int)blockDim.x >> 6; // wavefronts per workgroup: 4, fewer when four slice images do not fit in LDS
  extern __shared__ unsigned lds_u[];
  __shared__ unsigned long long s_base; // byte offset of this tile inside the picture payload
  __shared__ int s_tile, s_tot[4];
  PACK_STAMP(0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = lane / W, sl = lane % W; // slice of the wavefront, lane inside the slice
  const int pic = blockIdx.y;
  int tile = blockIdx.x;
  if (p.lookback) { // tiles are numbered in the order workgroups start, so a predecessor is always running
    if (threadIdx.x == 0) s_tile = (int)atomicAdd(p.lookback + (size_t)pic * p.lookback_stride, 1ull);
    __syncthreads();
    tile = s_tile;
  }
  const int slice = (tile * nwv + wave) * S + seg;
  const int img_words = (p.prefix + 4 + 3 * 255 * p.scalar + 3) / 4 + 2;
  const bool active = slice < p.n_slices;
  unsigned *img = GIMG ? (unsigned *)(p.slots + ((size_t)pic * p.n_slices + (active ? slice : 0)) * p.slot_bytes)
                       : lds_u + (wave * S + seg) * img_words;
  unsigned *lut = lds_u + (GIMG ? 0 : nwv * S * img_words);
  unsigned char *band_y = (unsigned char *)(lut + VLC_LUT_N), *band_c = band_y + 512;
  uint4 *qtab = (uint4 *)(band_c + 256) + (wave * S + seg) * 32;
  // slices of up to 2048 coefficients per component (more than one round of 512): the same tables, longer (launcher: big_lut)
  unsigned char *big_y = (unsigned char *)((uint4 *)(band_c + 256) + nwv * S * 32), *big_c = big_y + 2048;
  const bool fast = p.comp_n[0] <= 8 * W && p.comp_n[1] <= 4 * W && p.comp_n[1] == p.comp_n[2];
  // the luma round's loads go out before the tables: their latency and the tables' side by side (0.383 -> 0.372 ms; the
  // chroma round's as well: 81 registers, a wavefront per SIMD fewer, 0.40 ms)
  uint4 pre_y = make_uint4(0u, 0u, 0u, 0u);
  if constexpr (St<ST>::narrow && !GIMG) {
    if (fast && p.quantise && active && sl * 8 + 8 <= p.comp_n[0])
      pre_y = *(const uint4 *)((const ST *)p.store + (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs + p.comp_off[0] + sl * 8);
  }
  if (!GIMG || active) for (int i = sl; i < img_words; i += W) img[i] = 0;
  if (GIMG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the zeros are in place before the first atomic OR
  build_vlc_lut(lut);
  if (fast && p.quantise) copy_band_lut(band_y, p.band_lut);
  const bool mid = MID && W == 64 && !fast && p.quantise && p.big_lut; // its own instantiation: the one-round kernel keeps its 71 registers
  if (mid) {
    const int n0y = p.comp_n0[0], n0c = p.comp_n0[1];
    const int sy = (n0y & (n0y - 1)) == 0 ? 31 - __clz(n0y) : -1, sc = (n0c & (n0c - 1)) == 0 ? 31 - __clz(n0c) : -1;
    for (int j = threadIdx.x; j < 2048; j += blockDim.x) {
      big_y[j] = (unsigned char)band_of_index_fast(min(j, p.comp_n[0] - 1), n0y, sy);
      big_c[j] = (unsigned char)band_of_index_fast(min(j, p.comp_n[1] - 1), n0c, sc);
    }
  }
  if (p.quantise) {
    if (active && sl < 3 * p.depth + 1) { // quantiser constants of every subband for this slice's index
      const int aq = max(p.qidx[(size_t)pic * p.n_slices + slice] - p.qmatrix[sl], 0);
      // (magic, shift, factor, rounded-up 4 / factor as a float)
      if (aq > 119) { atomicOr(p.err, VC2_DEVERR_QINDEX); qtab[sl] = make_uint4(0u, 0u, 0x40000000u, 0u); }
      else qtab[sl] = make_uint4(c_qs.magic[aq], (unsigned)c_qs.shift[aq], (unsigned)c_qs.qf[aq], __float_as_uint(c_qs.inv4[aq]));
    }
  }
  __syncthreads();
  PACK_STAMP(1);

  int bytes[3] = {0, 0, 0};
  int q = 0;
  bool bad_cbr = false;
  if (active) {
    q = p.qidx[(size_t)pic * p.n_slices + slice];
    const size_t rec_at = (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs;
    const ST *rec = (const ST *)p.store + rec_at;
    const int32_t *recw = St<ST>::narrow ? p.store_wide + rec_at : nullptr;
    const int cbr_total = p.cbr_bytes ? p.cbr_bytes[slice] : 0;
    auto comp_len = [&](int count) -> int { // ceil(bytes / scalar) by the rounded-up reciprocal (exact far beyond 255 * scalar)
      int len = (int)((float)(((count + 7) >> 3) + p.scalar - 1) * p.inv_scalar);
      if (len > 255) { if (sl == 0) atomicOr(p.err, VC2_DEVERR_SCALAR); len = 255; }
      return len * p.scalar;
    };
    auto cbr_v = [&](int need) -> int { // Slices.cpp:352-368: V absorbs the remainder of the slice
      if (!p.cbr_bytes) return need;
      const int vb = cbr_total - 4 - bytes[0] - bytes[1];
      if (vb < need) { if (sl == 0) atomicOr(p.err, VC2_DEVERR_CBR_TOOBIG); bad_cbr = true; return need; }
      if (vb / p.scalar > 255) { if (sl == 0) atomicOr(p.err, VC2_DEVERR_CBR_LEN); bad_cbr = true; return need; }
      return vb;
    };
    int base = p.prefix + 1; // byte offset of the next component's length byte
    if (fast) {
      Coef8 c;
      { // luma: one round
        const int n = p.comp_n[0], n0 = p.comp_n0[0];
        const int n0s = (n0 & (n0 - 1)) == 0 ? 31 - __clz(n0) : -1;
        if (p.quantise) load8_tab<ST, !GIMG>(c, rec + p.comp_off[0], recw + p.comp_off[0], sl * 8, n, band_y, qtab, p.err, lut, pre_y);
        else load8<false>(c, rec + p.comp_off[0], recw + p.comp_off[0], sl * 8, n, n0, n0s, qtab, p.err, lut);
        PACK_STAMP(6);
        const int incl = seg_incl_scan<W>(c.sum, sl);
        const int count = seg_max<W>(c.last_end ? incl - c.sum + c.last_end : 0);
        bytes[0] = comp_len(count);
        PACK_STAMP(7);
        const int lim_y = skip_codes ? 0 : 8 * (base + 1 + bytes[0]);
        if (!__any(c.sum > 64)) write8_short(img, 8 * (base + 1) + incl - c.sum, lim_y, c);
        else write8(img, 8 * (base + 1) + incl - c.sum, lim_y, c);
        if (sl == 0) put_byte(img, base, (unsigned)(bytes[0] / p.scalar));
        base += 1 + bytes[0];
      }
      PACK_STAMP(2);
      { // both chroma components in one round: the lower half of the lanes U, the upper half V
        const int n = p.comp_n[1], n0 = p.comp_n0[1];
        const int n0s = (n0 & (n0 - 1)) == 0 ? 31 - __clz(n0) : -1;
        const int half = sl >= W / 2 ? 1 : 0, cc = 1 + half;
        if (p.quantise) load8_tab(c, rec + p.comp_off[cc], recw + p.comp_off[cc], (sl & (W / 2 - 1)) * 8, n, band_c, qtab, p.err, lut);
        else load8<false>(c, rec + p.comp_off[cc], recw + p.comp_off[cc], (sl & (W / 2 - 1)) * 8, n, n0, n0s, qtab, p.err, lut);
        const int incl = seg_incl_scan<W>(c.sum, sl);
        const int total_u = __shfl(incl, seg * W + W / 2 - 1);
        const int rel = incl - c.sum - (half ? total_u : 0);
        const int cnt = seg_max<W / 2>(c.last_end ? rel + c.last_end : 0);
        bytes[1] = comp_len(__shfl(cnt, seg * W));
        bytes[2] = cbr_v(comp_len(__shfl(cnt, seg * W + W / 2)));
        const int base_c = half ? base + 1 + bytes[1] : base;
        const int lim_c = skip_codes ? 0 : 8 * (base_c + 1 + bytes[cc]);
        if (!__any(c.sum > 64)) write8_short(img, 8 * (base_c + 1) + rel, lim_c, c);
        else write8(img, 8 * (base_c + 1) + rel, lim_c, c);
        if (sl == 0) put_byte(img, base, (unsigned)(bytes[1] / p.scalar));
        if (sl == W / 2) put_byte(img, base_c, (unsigned)(bytes[2] / p.scalar));
      }
    } else if constexpr (W == 64) { // any geometry: two passes per component (measure, then write), 512 coefficients per round
      for (int cc = 0; cc < 3; ++cc) {
        const int n = p.comp_n[cc], n0 = p.comp_n0[cc];
        const int n0s = (n0 & (n0 - 1)) == 0 ? 31 - __clz(n0) : -1;
        const ST *src = rec + p.comp_off[cc];
        const int32_t *srcw = recw + p.comp_off[cc];
        Coef8 c;
        int run = 0, count = 0;
        const unsigned char *band_lut = cc ? big_c : big_y;
        if constexpr (MID) {
          if (mid && n <= 1024) { // at most two rounds: codes of both kept in registers, quantised once (no measuring pass)
            Coef8 c1;
            load8_tab(c, src, srcw, lane * 8, n, band_lut, qtab, p.err, lut);
            const bool two = n > 512;
            if (two) load8_tab(c1, src, srcw, 512 + lane * 8, n, band_lut, qtab, p.err, lut);
            else {
              c1.sum = 0; c1.last_end = 0;
#pragma unroll
              for (int k = 0; k < 8; ++k) { c1.nb[k] = 0; c1.code[k] = 0; }
            }
            const int incl0 = wave_incl_scan(c.sum, lane), run0 = __builtin_amdgcn_readlane(incl0, 63);
            const int incl1 = wave_incl_scan(c1.sum, lane);
            count = max(wave_max(c.last_end ? incl0 - c.sum + c.last_end : 0),
                        wave_max(c1.last_end ? run0 + incl1 - c1.sum + c1.last_end : 0));
            bytes[cc] = comp_len(count);
            if (cc == 2) bytes[2] = cbr_v(bytes[2]);
            write8(img, 8 * (base + 1) + incl0 - c.sum, 8 * (base + 1 + bytes[cc]), c);
            if (two) write8(img, 8 * (base + 1) + run0 + incl1 - c1.sum, 8 * (base + 1 + bytes[cc]), c1);
            if (lane == 0) put_byte(img, base, (unsigned)(bytes[cc] / p.scalar));
            base += 1 + bytes[cc];
            continue;
          }
        }
        for (int r0 = 0; r0 < n; r0 += 512) {
          if (mid) bits8_tab(src, srcw, r0 + lane * 8, n, band_lut, qtab, p.err, c.sum, c.last_end); // lengths only
          else if (p.quantise) load8<true>(c, src, srcw, r0 + lane * 8, n, n0, n0s, qtab, p.err, lut);
          else load8<false>(c, src, srcw, r0 + lane * 8, n, n0, n0s, qtab, p.err, lut);
          const int incl = wave_incl_scan(c.sum, lane);
          count = max(count, wave_max(c.last_end ? run + incl - c.sum + c.last_end : 0));
          run += __builtin_amdgcn_readlane(incl, 63);
        }
        bytes[cc] = comp_len(count);
        if (cc == 2) bytes[2] = cbr_v(bytes[2]);
        run = 0;
        for (int r0 = 0; r0 < n; r0 += 512) {
          if (mid) load8_tab(c, src, srcw, r0 + lane * 8, n, band_lut, qtab, p.err, lut);
          else if (p.quantise) load8<true>(c, src, srcw, r0 + lane * 8, n, n0, n0s, qtab, p.err, lut);
          else load8<false>(c, src, srcw, r0 + lane * 8, n, n0, n0s, qtab, p.err, lut);
          const int incl = wave_incl_scan(c.sum, lane);
          write8(img, 8 * (base + 1) + run + incl - c.sum, 8 * (base + 1 + bytes[cc]), c);
          run += __builtin_amdgcn_readlane(incl, 63);
        }
        if (lane == 0) put_byte(img, base, (unsigned)(bytes[cc] / p.scalar));
        base += 1 + bytes[cc];
      }
    }
    if (sl == 0) put_byte(img, p.prefix, (unsigned)q & 0xFF);
  }
  const int total = active ? p.prefix + 4 + bytes[0] + bytes[1] + bytes[2] : 0;
  // tile_slices: the slices of the workgroup go into its slot back to back (the wavefronts exchange their byte counts over
  // the barrier below), one size per workgroup: the compaction then moves one run of ~1.2 KB per workgroup instead of
  // four of ~290 bytes
  // (only the instantiations with several slices per wavefront: the caller shares slots only there, vc2hip_api.hip)
  const bool tiled = S > 1 && p.tile_slices;
  int wave_total = total, seg_off = 0; // bytes of the wavefront's slices; of those before this lane's slice
  if constexpr (S > 1) {
    wave_total = 0;
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) { const int t = __shfl(total, s2 * W); wave_total += t; if (s2 < seg) seg_off += t; }
  }
  if ((p.lookback || tiled) && lane == 0) s_tot[wave] = wave_total;
  PACK_STAMP(3);
  __syncthreads();
  PACK_STAMP(4);
  if (tiled && threadIdx.x == 0) {
    unsigned all = 0;
    for (int w2 = 0; w2 < nwv; ++w2) all += (unsigned)s_tot[w2];
    p.sizes[(size_t)pic * gridDim.x + tile] = all;
  }
  if (p.lookback) {
    // decoupled look-back: publish this tile's byte count, add up the predecessors' counts until one
    // of them carries an inclusive prefix, publish ours.  One 8-byte agent-scope word per tile holds
    // flag and value together, so no other ordering is needed.
    if (wave == 0) { // the whole first wavefront looks back, 64 predecessor tiles per step
      unsigned long long *st = p.lookback + (size_t)pic * p.lookback_stride + 1;
      const unsigned long long agg = (unsigned long long)(s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3]);
      const unsigned long long M62 = (1ull << 62) - 1;
      unsigned long long run = 0;
      if (tile > 0) {
        if (lane == 0) __hip_atomic_store(&st[tile], (1ull << 62) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool done = false;
        for (int base = tile - 1; !done; base -= 64) {
          const int t = base - lane;
          for (int spins = 0;; ++spins) {
            // tiles before the first one count as an inclusive prefix of zero
            const unsigned long long v = t >= 0 ? __hip_atomic_load(&st[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
            const unsigned flag = (unsigned)(v >> 62);
            const unsigned long long m2 = __ballot(flag == 2), m0 = __ballot(flag == 0);
            const int first2 = m2 ? __ffsll((long long)m2) - 1 : 64;
            const unsigned long long need = first2 >= 63 ? ~0ull : ((2ull << first2) - 1);
            if ((m0 & need) == 0) { // every tile up to the first inclusive prefix has published
              run += (unsigned long long)wave_sum64((long long)(lane <= first2 ? (v & M62) : 0ull));
              done = first2 < 64;
              break;
            }
            __builtin_amdgcn_s_sleep(2);
            if (spins > (1 << 22)) { if (lane == 0) atomicOr(p.err, VC2_DEVERR_STREAM); done = true; break; }
          }
        }
      }
      if (lane == 0) {
        __hip_atomic_store(&st[tile], (2ull << 62) | (run + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_base = run;
        if (tile == (p.n_slices + 3) / 4 - 1) p.lens[pic] = run + agg;
      }
    }
    __syncthreads();
  }
  if (!active || skip_out) return;

  if (p.lookback) {
    unsigned long long off = s_base;
    for (int w2 = 0; w2 < wave; ++w2) off += (unsigned long long)s_tot[w2];
    uint8_t *dst = p.payload + (size_t)pic * p.payload_stride + off;
    // bytes of the image in stream order: byte i = img[i >> 2] >> (24 - 8 * (i & 3)); dword stores for the
    // 4-byte aligned middle of the destination, byte stores for its ragged head and tail
    const int head = min((int)((4 - ((size_t)dst & 3)) & 3), total);
    const int nw = (total - head) >> 2, tail0 = head + 4 * nw;
    if (lane < head) dst[lane] = (uint8_t)(img[lane >> 2] >> (24 - 8 * (lane & 3)));
    if (lane < total - tail0) { const int i = tail0 + lane; dst[i] = (uint8_t)(img[i >> 2] >> (24 - 8 * (i & 3))); }
    unsigned *d4 = (unsigned *)(dst + head);
    for (int w = lane; w < nw; w += 64) {
      const int i0 = head + 4 * w;
      const unsigned lo = __builtin_bswap32(img[i0 >> 2]), hi = __builtin_bswap32(img[(i0 >> 2) + 1]);
      d4[w] = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(i0 & 3));
    }
    return;
  }
  if (GIMG) { // the image is the slot: big-endian words to stream order in place, then the size
    if (bad_cbr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = sl; i < (total + 3) / 4; i += W) {
      const unsigned w = __hip_atomic_load(&img[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&img[i], __builtin_bswap32(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (sl == 0) p.sizes[(size_t)pic * p.n_slices + slice] = (unsigned)total;
  } else if (p.cbr_bytes) {
    if (bad_cbr) return;
    uint8_t *dst = p.payload + (size_t)pic * p.payload_stride + p.cbr_offsets[slice];
    // dword stores for the 4-byte aligned middle of the destination (the budgets put a slice at any byte), byte stores for
    // its ragged head and tail (byte stores for all of it: cfg 3's pack 0.43 ms against 0.385 ms in VBR mode)
    const int head = min((int)((4 - ((size_t)dst & 3)) & 3), total);
    const int nw = (total - head) >> 2, tail0 = head + 4 * nw;
    if (sl < head) dst[sl] = (uint8_t)(img[sl >> 2] >> (24 - 8 * (sl & 3)));
    if (sl < total - tail0) { const int i = tail0 + sl; dst[i] = (uint8_t)(img[i >> 2] >> (24 - 8 * (i & 3))); }
    unsigned *d4 = (unsigned *)(dst + head);
    for (int w = sl; w < nw; w += W) {
      const int i0 = head + 4 * w;
      const unsigned lo = __builtin_bswap32(img[i0 >> 2]), hi = __builtin_bswap32(img[(i0 >> 2) + 1]);
      d4[w] = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(i0 & 3));
    }
  } else if (tiled) {
    int off = seg_off;
    for (int w2 = 0; w2 < wave; ++w2) off += s_tot[w2];
    uint8_t *dst = p.slots + ((size_t)pic * gridDim.x + tile) * ((size_t)p.tile_slices * p.slot_bytes) + off;
    const int head = min((int)((4 - ((size_t)dst & 3)) & 3), total);
    const int nw = (total - head) >> 2, tail0 = head + 4 * nw;
    if (sl < head) dst[sl] = (uint8_t)(img[sl >> 2] >> (24 - 8 * (sl & 3)));
    if (sl < total - tail0) { const int i = tail0 + sl; dst[i] = (uint8_t)(img[i >> 2] >> (24 - 8 * (i & 3))); }
    unsigned *d4 = (unsigned *)(dst + head);
    for (int w = sl; w < nw; w += W) {
      const int i0 = head + 4 * w;
      const unsigned lo = __builtin_bswap32(img[i0 >> 2]), hi = __builtin_bswap32(img[(i0 >> 2) + 1]);
      d4[w] = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(i0 & 3));
    }
  } else {
    unsigned *dst = (unsigned *)(p.slots + ((size_t)pic * p.n_slices + slice) * p.slot_bytes);
    if (sl == 0) p.sizes[(size_t)pic * p.n_slices + slice] = (unsigned)total;
    for (int i = sl; i < (total + 3) / 4; i += W) dst[i] = __builtin_bswap32(img[i]);
  }
  PACK_STAMP(5);
}

struct __attribute__((aligned(4))) Dword4 { unsigned x, y, z, w; }; // four dwords at any dword-aligned address
#include "vc2hip_pack16.h"

size_t vc2_pack_lds_bytes(int prefix, int scalar);
static size_t pack_lds(int prefix, int scalar, int slices_per_wave, bool big_lut = false, int waves = 4, bool gimg = false) {
  const size_t img_words = gimg ? 0 : ((size_t)prefix + 4 + 3 * 255 * (size_t)scalar + 3) / 4 + 2;
  return waves * slices_per_wave * img_words * 4 + VLC_LUT_N * 4 + 768 + waves * (size_t)slices_per_wave * 32 * 16 + (big_lut ? 4096 : 0);
}
// 0: four slice images fit in LDS; else the wavefronts per workgroup to use (2, 1), or -1: images in global memory
int vc2_pack_image_mode(int prefix, int scalar) {
  if (pack_lds(prefix, scalar, 1) <= 144 * 1024) return 0;
  if (pack_lds(prefix, scalar, 1, false, 2) <= 144 * 1024) return 2;
  if (pack_lds(prefix, scalar, 1, false, 1) <= 144 * 1024) return 1;
  return -1;
}
size_t vc2_pack_lds_bytes(int prefix, int scalar) { return pack_lds(prefix, scalar, 1); }
// lanes per slice: as few as still hold the slice (8 luma / 4 chroma coefficients and one subband constant per lane)
static int pack_lanes(const PackParams &p) {
  int W = 64;
  const bool same_c = p.comp_n[1] == p.comp_n[2];
  static const int force_w = vc2_tune_int("VC2HIP_PACK_LANES", 0);
  if (!p.lookback && same_c && force_w != 64) {
    const int bands = 3 * p.depth + 1;
    if (p.comp_n[0] <= 128 && p.comp_n[1] <= 64 && bands <= 16 && pack_lds(p.prefix, p.scalar, 4) <= 144 * 1024) W = 16;
    else if (p.comp_n[0] <= 256 && p.comp_n[1] <= 128 && bands <= 32 && pack_lds(p.prefix, p.scalar, 2) <= 144 * 1024) W = 32;
  }
  return W;
}
// the consecutive slices of a pack workgroup that share a slot (PackParams::tile_slices), 0: one slot per slice -- the
// choice of the caller's comment (vc2hip_api.hip): shared slots where a wavefront packs two or four slices
int vc2_pack_slices_per_tile(const PackParams &p) {
  if (vc2_pack_image_mode(p.prefix, p.scalar) != 0) return 0;
  const int W = pack_lanes(p);
  return W == 64 ? 0 : 4 * (64 / W);
}
#ifdef VC2HIP_ABLATE
static void p16_print_stats(hipStream_t s) {
  static const bool on = getenv("VC2HIP_P16_STATS") != nullptr;
  if (!on) return;
  unsigned h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)hipStreamSynchronize(s);
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_p16_stats), sizeof h);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_p16_stats), z, sizeof z);
  fprintf(stderr, "pack16: wavefronts %u, store escape %u, quotient beyond the table %u, string beyond 63 bits %u, general path %u\n", h[0], h[1], h[2], h[3], h[4]);
}
#else
static void p16_print_stats(hipStream_t) {}
#endif
// VBR pictures whose slices k_hq_pack16 codes are packed in ONE pass by default (look-back, no slots): the caller asks
// before it allocates slots or look-back words
bool vc2_pack_one_pass_default(const PackParams &p0) {
  static const int use16 = vc2_tune_int("VC2HIP_PACK16", 1);
  if (!use16 || p0.cbr_bytes || vc2_pack_image_mode(p0.prefix, p0.scalar) != 0) return false;
  PackParams p = p0;
  static unsigned long long dummy;
  p.lookback = &dummy;
  p.tile_slices = 0;
  return pack16_plan(p, p.lane16);
}
void vc2_launch_pack(Launcher &L, const PackParams &p0, int n_pictures, hipStream_t s) {
  PackParams p = p0;
#ifdef VC2HIP_ABLATE
  { const char *e = getenv("VC2HIP_DEBUG_PACK"); p.debug_skip = (e ? atoi(e) : 0) | (getenv("VC2HIP_P16_STATS") ? 8 : 0); }
#endif
  p.inv_scalar = 1.0f / (float)p.scalar; // the smallest float >= 1 / scalar
  if ((double)p.inv_scalar < 1.0 / (double)p.scalar) p.inv_scalar = nextafterf(p.inv_scalar, INFINITY);
  static const int use16 = vc2_tune_int("VC2HIP_PACK16", 1);
  if (use16 && vc2_pack_image_mode(p.prefix, p.scalar) == 0 && pack16_plan(p, p.lane16)) {
    // the common geometry on the 16-bit store: one round per slice, sixteen coefficients per lane (vc2hip_pack16.h)
    const size_t lds = pack16_lds(p.prefix, p.scalar);
    const dim3 grid((p.n_slices + 3) / 4, n_pictures);
    vc2_prof_begin(L, "hq_pack", s);
    if (p.cbr_bytes) {
      vc2_allow_lds((const void *)k_hq_pack16<P16_CBR>, 144 * 1024);
      VC2_LAUNCH(L, k_hq_pack16<P16_CBR>, grid, dim3(256), lds, s, p);
    } else if (p.lookback) { // one pass: (pictures, tiles) -- see the kernel
      vc2_allow_lds((const void *)k_hq_pack16<P16_LOOKBACK>, 144 * 1024);
      VC2_LAUNCH(L, k_hq_pack16<P16_LOOKBACK>, dim3(n_pictures, (p.n_slices + P16LB_WAVES - 1) / P16LB_WAVES), dim3(64 * P16LB_WAVES),
                 pack16_lds(p.prefix, p.scalar, P16LB_WAVES), s, p);
    } else {
      vc2_allow_lds((const void *)k_hq_pack16<P16_SLOTS>, 144 * 1024);
      VC2_LAUNCH(L, k_hq_pack16<P16_SLOTS>, grid, dim3(256), lds, s, p);
    }
    vc2_prof_end(L, s);
    p16_print_stats(s);
    return;
  }
  if (use16 && pack16w_lds(p.prefix, p.scalar) <= 64 * 1024 && pack16w_plan(p, p.lane16)) {
    // large slices on the 16-bit store: a wavefront per component, a workgroup per slice (vc2hip_pack16.h)
    const size_t lds = pack16w_lds(p.prefix, p.scalar);
    const dim3 grid(p.n_slices, n_pictures);
    vc2_prof_begin(L, "hq_pack", s);
    if (p.cbr_bytes) {
      vc2_allow_lds((const void *)k_hq_pack16w<true>, 64 * 1024);
      VC2_LAUNCH(L, k_hq_pack16w<true>, grid, dim3(192), lds, s, p);
    } else {
      vc2_allow_lds((const void *)k_hq_pack16w<false>, 64 * 1024);
      VC2_LAUNCH(L, k_hq_pack16w<false>, grid, dim3(192), lds, s, p);
    }
    vc2_prof_end(L, s);
    p16_print_stats(s);
    return;
  }
  fill_band_lut(p.band_lut, p.comp_n, p.comp_n0);
  const int W = pack_lanes(p);
  const bool same_c = p.comp_n[1] == p.comp_n[2];
  const int S = 64 / W;
  const bool one_round = p.comp_n[0] <= 512 && p.comp_n[1] <= 256 && same_c;
  p.big_lut = W == 64 && !one_round && same_c && p.comp_n[0] <= 2048 && p.comp_n[1] <= 2048 && 3 * p.depth + 1 <= 32 &&
              pack_lds(p.prefix, p.scalar, 1, true) <= 144 * 1024;
  const int mode = vc2_pack_image_mode(p.prefix, p.scalar);
  if (mode != 0) { // long slices: fewer wavefronts per workgroup, or the images in the slots in global memory
    const int waves = mode < 0 ? 4 : mode;
    const size_t lds = pack_lds(p.prefix, p.scalar, 1, false, waves, mode < 0);
    const int tiles = (p.n_slices + waves - 1) / waves;
    vc2_prof_begin(L, "hq_pack", s);
#define VC2_PACK_LONG(TT, GG)                                                                         \
  do {                                                                                                \
    vc2_allow_lds((const void *)k_hq_pack<64, false, TT, GG>, 144 * 1024);                            \
    VC2_LAUNCH(L, (k_hq_pack<64, false, TT, GG>), dim3(tiles, n_pictures), dim3(64 * waves), lds, s, p); \
  } while (0)
    if (mode < 0) { if (p.store16) VC2_PACK_LONG(int16_t, true); else VC2_PACK_LONG(int32_t, true); }
    else { if (p.store16) VC2_PACK_LONG(int16_t, false); else VC2_PACK_LONG(int32_t, false); }
#undef VC2_PACK_LONG
    vc2_prof_end(L, s);
    return;
  }
  const size_t lds = pack_lds(p.prefix, p.scalar, S, p.big_lut);
  const int tiles = (p.n_slices + 4 * S - 1) / (4 * S);
  vc2_prof_begin(L, "hq_pack", s);
#define VC2_PACK_LAUNCH(WW, MM, TT)                                                                   \
  do {                                                                                                \
    vc2_allow_lds((const void *)k_hq_pack<WW, MM, TT>, 144 * 1024);                                   \
    VC2_LAUNCH(L, (k_hq_pack<WW, MM, TT>), dim3(tiles, n_pictures), dim3(256), lds, s, p);            \
  } while (0)
#ifdef VC2HIP_STAMPS
  const char *stamp_file = getenv("VC2HIP_PACK_STAMPS_FILE");
  const size_t stamp_n = (size_t)tiles * 8;
  unsigned long long *d_st = nullptr;
  if (stamp_file) {
    (void)hipMalloc((void **)&d_st, stamp_n * 8);
    (void)hipMemsetAsync(d_st, 0, stamp_n * 8, s);
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_pack_stamps), &d_st, sizeof d_st, 0, hipMemcpyHostToDevice, s);
  }
#endif
  if (p.store16) {
    if (W == 16) VC2_PACK_LAUNCH(16, false, int16_t);
    else if (W == 32) VC2_PACK_LAUNCH(32, false, int16_t);
    else if (p.big_lut) VC2_PACK_LAUNCH(64, true, int16_t);
    else VC2_PACK_LAUNCH(64, false, int16_t);
  } else {
    if (W == 16) VC2_PACK_LAUNCH(16, false, int32_t);
    else if (W == 32) VC2_PACK_LAUNCH(32, false, int32_t);
    else if (p.big_lut) VC2_PACK_LAUNCH(64, true, int32_t);
    else VC2_PACK_LAUNCH(64, false, int32_t);
  }
#undef VC2_PACK_LAUNCH
#ifdef VC2HIP_STAMPS
  if (stamp_file) {
    std::vector<unsigned long long> h(stamp_n);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), d_st, stamp_n * 8, hipMemcpyDeviceToHost);
    unsigned long long *none = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pack_stamps), &none, sizeof none);
    (void)hipFree(d_st);
    if (FILE *fp = fopen(stamp_file, "wb")) { fwrite(h.data(), 8, stamp_n, fp); fclose(fp); }
  }
#endif
  vc2_prof_end(L, s);
}

// ------------------------------------------------------------------------------------------
// VBR: exclusive scan of slice sizes (one workgroup per picture) + compaction
// ------------------------------------------------------------------------------------------
// PASSES > 0: a wavefront owns a contiguous sixteenth of the picture's sizes and takes it in PASSES passes of 64 -- every
// load and store coalesced, all loads in flight at once (n <= 1024 * PASSES).  PASSES == 0: any n, a thread sums its own
// contiguous run (one load after the other).  Both: a scan inside every wavefront (DPP) and one of the sixteen wavefront
// totals, two barriers.  (16 UHD pictures: 20 us with per-thread runs and a Hillis-Steele scan of 1024 partial sums
// through LDS, 5 us like this.)
template <int PASSES>
__global__ __launch_bounds__(1024) void k_scan_sizes(const uint32_t *sizes, uint32_t *offsets,
                                                     unsigned long long *totals, int n) {
  __shared__ unsigned wtot[16];
  const int pic = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t *s = sizes + (size_t)pic * n;
  uint32_t *o = offsets + (size_t)pic * n;
  auto wave_totals = [&](unsigned mine) -> unsigned { // exclusive scan over the wavefronts' totals
    if (lane == 63) wtot[wave] = mine;
    __syncthreads();
    if (wave == 0) {
      const unsigned v = lane < 16 ? wtot[lane] : 0u;
      const unsigned iv = (unsigned)wave_incl_scan((int)v, lane);
      if (lane < 16) wtot[lane] = iv - v;
      if (lane == 15) totals[pic] = iv;
    }
    __syncthreads();
    return wtot[wave];
  };
  if constexpr (PASSES > 0) {
    const int chunk = (n + 15) / 16, c0 = wave * chunk, c1 = min(n, c0 + chunk);
    unsigned v[PASSES], ex[PASSES];
#pragma unroll
    for (int k = 0; k < PASSES; ++k) { const int i = c0 + 64 * k + lane; v[k] = i < c1 ? s[i] : 0u; }
    unsigned carry = 0;
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const unsigned incl = (unsigned)wave_incl_scan((int)v[k], lane);
      ex[k] = carry + incl - v[k];
      carry += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    }
    const unsigned base = wave_totals(carry); // (carry is the same in every lane)
#pragma unroll
    for (int k = 0; k < PASSES; ++k) { const int i = c0 + 64 * k + lane; if (i < c1) o[i] = base + ex[k]; }
  } else {
    const int per = (n + 1023) / 1024;
    const int b = t * per, e = min(n, b + per);
    unsigned sum = 0;
    for (int i = b; i < e; ++i) sum += s[i];
    const unsigned incl = (unsigned)wave_incl_scan((int)sum, lane);
    unsigned run = wave_totals(incl) + incl - sum;
    for (int i = b; i < e; ++i) { o[i] = run; run += s[i]; }
  }
}

void vc2_launch_scan_sizes(Launcher &L, const uint32_t *sizes, uint32_t *offsets,
                           unsigned long long *totals, int n_slices, int n_pictures, hipStream_t s) {
  vc2_prof_begin(L, "slice_offsets_scan", s);
  if (n_slices <= 1024 * 16) VC2_LAUNCH(L, k_scan_sizes<16>, dim3(n_pictures), dim3(1024), 0, s, sizes, offsets, totals, n_slices);
  else if (n_slices <= 1024 * 32) VC2_LAUNCH(L, k_scan_sizes<32>, dim3(n_pictures), dim3(1024), 0, s, sizes, offsets, totals, n_slices);
  else VC2_LAUNCH(L, k_scan_sizes<0>, dim3(n_pictures), dim3(1024), 0, s, sizes, offsets, totals, n_slices);
  vc2_prof_end(L, s);
}

// W lanes copy one slice: 64, or 32 / 16 when slices are short (at most 4 * W dwords per trip would leave lanes idle)
template <int W>
__global__ __launch_bounds__(256) void k_compact(const uint8_t *slots, int slot_bytes,
                                                 const uint32_t *sizes, const uint32_t *offsets,
                                                 uint8_t *payload, long long payload_stride, int n) {
  const int sl = threadIdx.x % W, slice = (blockIdx.x * 256 + threadIdx.x) / W, pic = blockIdx.y;
  if (slice >= n) return;
  const size_t si = (size_t)pic * n + slice;
  const uint8_t *src = slots + si * slot_bytes; // 16-byte aligned
  // The first two trips' source pieces are requested BEFORE the slice's size and offset are known (the slot is the
  // slice's own memory up to slot_bytes whatever its size): the kernel is a chain of memory latencies -- size / offset,
  // then the source, then the stores (rocprofv3: 93 % of a wavefront's life waiting) -- and this puts the first two side
  // by side.  A typical UHD slice (~580 bytes) is those two trips of 32 lanes.
  constexpr int PRE = 2; // trips requested up front (16 lanes: 512 bytes, 32: 1024; three trips of 16 lanes: 0.132 against 0.117 ms)
  const bool pre = PRE * W * 16 + 16 <= slot_bytes;
  uint4 vp[PRE];
  unsigned nxp[PRE];
#pragma unroll
  for (int k = 0; k < PRE; ++k) { vp[k] = make_uint4(0u, 0u, 0u, 0u); nxp[k] = 0; }
  if (pre) {
    const unsigned *s4p = (const unsigned *)src;
#pragma unroll
    for (int k = 0; k < PRE; ++k) { vp[k] = *(const uint4 *)(s4p + 4 * (sl + k * W)); nxp[k] = s4p[4 * (sl + k * W) + 4]; }
  }
  uint8_t *dst = payload + (size_t)pic * payload_stride + offsets[si];
  const int size = (int)sizes[si];
  // dword stores for the 4-byte aligned middle of the destination (each built from two aligned source
  // dwords), byte stores for its ragged head and tail
  const int head = min((int)((4 - ((size_t)dst & 3)) & 3), size);
  const int nw = (size - head) >> 2, tail0 = head + 4 * nw;
  if (sl < head) dst[sl] = src[sl];
  if (sl < size - tail0) dst[tail0 + sl] = src[tail0 + sl];
  // sixteen bytes per lane and trip: destination dword w holds source bytes head + 4 w ... = source dwords w and w + 1
  // shifted by `head` bytes (head < 4), so a lane loads its four source dwords in one piece and the one behind them, and
  // stores four destination dwords in one piece (dword-aligned: the destination starts anywhere).  A typical slice of
  // 290 bytes is one load and one store instruction of its 32 lanes where a dword per lane and trip took three trips of
  // two loads and a store (0.098 -> 0.087 ms per 16 UHD pictures)
  const unsigned *s4 = (const unsigned *)src;
  unsigned *d4 = (unsigned *)(dst + head);
  for (int q = sl; 4 * q < nw; q += W) {
    uint4 v;
    unsigned nx;
    const int trip = (q - sl) / W;
    if (pre && trip < PRE) {
      v = vp[0]; nx = nxp[0];
#pragma unroll
      for (int k = 1; k < PRE; ++k) if (trip == k) { v = vp[k]; nx = nxp[k]; }
    } else { v = *(const uint4 *)(s4 + 4 * q); nx = s4[4 * q + 4]; }
    const unsigned h = (unsigned)head;
    Dword4 o;
    o.x = __builtin_amdgcn_alignbyte(v.y, v.x, h);
    o.y = __builtin_amdgcn_alignbyte(v.z, v.y, h);
    o.z = __builtin_amdgcn_alignbyte(v.w, v.z, h);
    o.w = __builtin_amdgcn_alignbyte(nx, v.w, h);
    if (4 * q + 4 <= nw) *(Dword4 *)(d4 + 4 * q) = o;
    else {
      d4[4 * q] = o.x;
      if (4 * q + 1 < nw) d4[4 * q + 1] = o.y;
      if (4 * q + 2 < nw) d4[4 * q + 2] = o.z;
    }
  }
}

void vc2_launch_compact(Launcher &L, const uint8_t *slots, int slot_bytes, const uint32_t *sizes,
                        const uint32_t *offsets, uint8_t *payload, long long payload_stride,
                        int n_slices, int n_pictures, hipStream_t s) {
  vc2_prof_begin(L, "slice_compact", s);
  // lanes per slice from the slot size (an upper bound of the slice size; typical slices are far shorter)
  static const int force = vc2_tune_int("VC2HIP_COMPACT_LANES", 0);
  // (round 4, with the first two trips' pieces requested up front: slots of one UHD slice, 1.5 KiB: 16 lanes 0.117 ms, 32 0.150,
  // 8 0.17, 64 0.19; of one UHD-2 slice, 6 KiB: 0.063 / 0.055 / - / 0.068; of sixteen HD slices, 12 KiB: 0.054 / 0.043 / 0.051)
  const int W = force ? force : (slot_bytes <= 2048 ? 16 : (slot_bytes <= 16384 ? 32 : 64));
  const int per_wg = 256 / W;
  const dim3 grid((n_slices + per_wg - 1) / per_wg, n_pictures);
  if (W == 8) VC2_LAUNCH(L, k_compact<8>, grid, dim3(256), 0, s, slots, slot_bytes, sizes, offsets, payload, payload_stride, n_slices);
  else if (W == 16) VC2_LAUNCH(L, k_compact<16>, grid, dim3(256), 0, s, slots, slot_bytes, sizes, offsets, payload, payload_stride, n_slices);
  else if (W == 32) VC2_LAUNCH(L, k_compact<32>, grid, dim3(256), 0, s, slots, slot_bytes, sizes, offsets, payload, payload_stride, n_slices);
  else VC2_LAUNCH(L, k_compact<64>, grid, dim3(256), 0, s, slots, slot_bytes, sizes, offsets, payload, payload_stride, n_slices);
  vc2_prof_end(L, s);
}

// bits of eight coefficients of one component, quantised through the tables: total and end of the last non-zero code
// (TRIAL: a quantiser search's measurement -- a code of more than 32 bits has its length and raises nothing, see component_bits)
template <class ST, bool TRIAL>
__device__ __forceinline__ void bits8_tab(const ST *src, const int32_t *wide, int j0, int n, const unsigned char *band_lut,
                                          const uint4 *qtab, unsigned *err, int &sum, int &last_end) {
  sum = 0; last_end = 0;
  if (j0 + 8 <= n) {
    int v[8];
    St<ST>::load8(src + j0, wide + j0, v);
    const uint2 bands = *(const uint2 *)(band_lut + j0);
    // the reciprocal multiply for all eight, one test whether any left its domain (see load8_tab)
    unsigned a8[8], qf8[8], dom = 0;
    int qq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint4 t = qtab[((k < 4 ? bands.x : bands.y) >> (8 * (k & 3))) & 0xFFu];
      const unsigned a = (v[k] < 0 ? 0u - (unsigned)v[k] : (unsigned)v[k]) << 2;
      const unsigned m = __umulhi(t.x, a);
      qq[k] = (int)((m + ((a - m) >> 1)) >> t.y);
      a8[k] = a; qf8[k] = t.z;
      dom |= a | (t.z - 2u);
    }
    if (__any((int)dom < 0)) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if ((int)(a8[k] | (qf8[k] - 2u)) < 0) qq[k] = (int)a8[k] / (int)qf8[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { // only the length matters: the sign of the quantised value does not change it
      const unsigned m1 = (unsigned)(qq[k] < 0 ? -qq[k] : qq[k]) + 1u;
      int nb = qq[k] == 0 ? 1 : 2 * (31 - __clz((int)m1)) + 2;
      if (!TRIAL && nb > 32) { atomicOr(err, VC2_DEVERR_CODE32); nb = 1; }
      sum += nb;
      if (qq[k] != 0) last_end = sum;
    }
  } else {
    for (int k = 0; k < 8 && j0 + k < n; ++k) {
      const uint4 t = qtab[band_lut[j0 + k]];
      const int c = quant_core(St<ST>::load1(src + j0 + k, wide + j0 + k), (int)t.z, t.x, (int)t.y);
      int nb = svlc_bits(c);
      if (!TRIAL && nb > 32) { atomicOr(err, VC2_DEVERR_CODE32); nb = 1; }
      sum += nb;
      if (c != 0) last_end = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------
// HQ_CBR quantiser search: one wavefront per slice, slice coefficients staged in LDS
// ------------------------------------------------------------------------------------------
// GLOBAL: the slice is too large to be staged in LDS (more than ~40 K coefficients): every trial reads it from the store
template <class ST, bool GLOBAL = false>
__global__ __launch_bounds__(256) void k_cbr_search(const CbrParams p) {
  extern __shared__ __attribute__((aligned(16))) int lds_i[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpw = blockDim.x >> 6; // 1..4 wavefronts per workgroup
  int slice = blockIdx.x * wpw + wave; const int pic = blockIdx.y;
  // Common geometry (component <= 512 / 256 coefficients): the subband of every coefficient index comes from a byte
  // table, the quantiser constants of every subband at the trial index from a per-wavefront table, both in LDS --
  // indexing the kernel-argument matrix and the constant-memory factor tables per lane costs several dependent
  // memory operations per coefficient.
  const bool fast = !GLOBAL && p.comp_n[0] <= 512 && p.comp_n[1] <= 256 && p.comp_n[1] == p.comp_n[2] && p.n_bands <= 32;
  unsigned char *band_y = (unsigned char *)(lds_i + (GLOBAL ? 0 : wpw * p.slice_coefs)), *band_c = band_y + 512;
  uint4 *qtab = (uint4 *)(band_c + 256) + wave * 32;
  if (fast) copy_band_lut(band_y, p.band_lut);
  __syncthreads();
  // no workgroup barriers below.  only_marked: the second pass behind k_cbr_search_reg -- a small grid walks the slices and
  // searches those the register kernel handed back
  auto search = [&](const int slice) {
  int *co = lds_i + (GLOBAL ? 0 : wave * p.slice_coefs);
  const size_t rec_at = (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs;
  const ST *rec = (const ST *)p.store + rec_at;
  const int32_t *recw = St<ST>::narrow ? p.store_wide + rec_at : nullptr;
  auto coef = [&](int i) -> int { // coefficient i of the slice record
    if constexpr (GLOBAL) return St<ST>::load1(rec + i, recw + i);
    else return co[i];
  };
  if constexpr (!GLOBAL) {
    if ((p.slice_coefs & 7) == 0 && (rec_at & 7) == 0) {
      for (int i = lane * 8; i < p.slice_coefs; i += 512) {
        int e[8];
        St<ST>::load8(rec + i, recw + i, e);
        *(int4 *)(co + i) = make_int4(e[0], e[1], e[2], e[3]);
        *(int4 *)(co + i + 4) = make_int4(e[4], e[5], e[6], e[7]);
      }
    } else {
      for (int i = lane; i < p.slice_coefs; i += 64) co[i] = St<ST>::load1(rec + i, recw + i);
    }
  }
  wave_lds_sync(); // no cross-wave sharing of `co`

  const int avail = p.slice_bytes[slice] - 4;
  // (magic, shift, factor, offset) of every subband at index tq; false if an adjusted index leaves the table
  auto set_q = [&](int tq) -> bool {
    bool ok = true;
    wave_lds_sync(); // the previous trial's readers are done
    if (lane < p.n_bands) {
      const int aq = max(tq - p.qmatrix[lane], 0);
      ok = aq <= 119;
      const int a = min(aq, 119);
      qtab[lane] = make_uint4(c_qs.magic[a], (unsigned)c_qs.shift[a], (unsigned)c_qs.qf[a], (unsigned)c_qs.off[a]);
    }
    wave_lds_sync();
    return !__any(!ok);
  };
  auto comp_bytes = [&](int count, bool &bad) -> int {
    const int len = ((count + 7) / 8 + p.scalar - 1) / p.scalar;
    if (len > 255) { bad = true; if (lane == 0) atomicOr(p.err, VC2_DEVERR_SCALAR); }
    return len * p.scalar;
  };
  auto need_bytes = [&](int tq, bool &bad) -> int {
    int need = 0;
    if (fast) {
      if (!set_q(tq)) { if (lane == 0) atomicOr(p.err, VC2_DEVERR_QINDEX); }
      int sum, last_end;
      bits8_tab<int32_t, true>(co + p.comp_off[0], nullptr, lane * 8, p.comp_n[0], band_y, qtab, p.err, sum, last_end); // luma: one round
      int incl = wave_incl_scan(sum, lane);
      need += comp_bytes(wave_max(last_end ? incl - sum + last_end : 0), bad);
      const int half = lane >> 5;                                               // lanes 0-31 U, lanes 32-63 V
      bits8_tab<int32_t, true>(co + p.comp_off[1 + half], nullptr, (lane & 31) * 8, p.comp_n[1], band_c, qtab, p.err, sum, last_end);
      incl = wave_incl_scan(sum, lane);
      const int total_u = __shfl(incl, 31); // every lane takes part in the shuffle
      const int rel = incl - sum - (half ? total_u : 0);
      const int cnt = seg_max<32>(last_end ? rel + last_end : 0);
      need += comp_bytes(__shfl(cnt, 0), bad) + comp_bytes(__shfl(cnt, 32), bad);
      return need;
    }
    for (int c = 0; c < 3; ++c) {
      if constexpr (GLOBAL)
        need += comp_bytes(component_bits<true, false>([&](int j) { return coef(p.comp_off[c] + j); }, p.comp_n[c], p.comp_n0[c], tq,
                                                       p.qmatrix, lane, nullptr, 0, p.err), bad);
      else need += comp_bytes(component_bits_lds(co + p.comp_off[c], p.comp_n[c], p.comp_n0[c], tq, p.qmatrix, lane, p.err), bad);
    }
    return need;
  };
  // luma-only sum of squared reconstruction error (int product, 64-bit sum)
  auto yss = [&](int tq, bool &bad) -> long long {
    long long acc = 0;
    const int *src = co + p.comp_off[0];
    if (fast) {
      if (!set_q(tq)) bad = true;
      for (int j = lane; j < p.comp_n[0]; j += 64) {
        const uint4 t = qtab[band_y[j]];
        const int v = src[j];
        const int qv = quant_core(v, (int)t.z, t.x, (int)t.y);
        // scale(), Quantisation.cpp:86-95, with the table's factor and offset
        const unsigned mag = qv < 0 ? 0u - (unsigned)qv : (unsigned)qv;
        int r = (int)(mag * t.z);
        if (r > 0) r = (int)((unsigned)r + t.w);
        r = (int)((unsigned)r + 2u);
        r /= 4;
        if (qv < 0) r = (int)(0u - (unsigned)r);
        const int d = (int)((unsigned)v - (unsigned)r);
        acc += (int)((unsigned)d * (unsigned)d);
      }
      return wave_sum64(acc);
    }
    for (int j = lane; j < p.comp_n[0]; j += 64) {
      const int aq = max(tq - p.qmatrix[band_of_index(j, p.comp_n0[0])], 0);
      if (aq > 119) { bad = true; continue; }
      const int v = GLOBAL ? coef(p.comp_off[0] + j) : src[j];
      const int d = (int)((unsigned)v - (unsigned)scale_dev(quant_dev(v, aq), aq));
      acc += (int)((unsigned)d * (unsigned)d);
    }
    return wave_sum64(acc);
  };

  bool bad = false;
  int trial = 63, q = 127, delta = 64;
  while (delta > 0) {
    delta >>= 1;
    const int need = need_bytes(trial, bad);
    bad = __any(bad);
    if (bad) break;
    if (need <= avail) { if (trial < q) q = trial; trial -= delta; }
    else trial += delta;
  }
  if (!bad) {
    trial = q;
    long long prev = yss(trial, bad), d;
    do {
      ++trial;
      const long long cur = yss(trial, bad);
      bad = __any(bad);
      if (bad) break;
      d = cur - prev;
      prev = cur;
    } while (d < 0);
    q = trial - 1;
  }
  if (__any(bad) && lane == 0) atomicOr(p.err, VC2_DEVERR_QINDEX);
  if (lane == 0) p.qidx[(size_t)pic * p.n_slices + slice] = q;
  wave_lds_sync();
  }; // search(slice)
  if (!p.only_marked) {
    if (slice < p.n_slices) search(slice);
    return;
  }
  // 64 indices per look: the wavefronts of the small grid stride over the picture's slices
  for (int base = slice * 64; base < p.n_slices; base += (int)gridDim.x * wpw * 64) {
    const bool marked = base + lane < p.n_slices && p.qidx[(size_t)pic * p.n_slices + base + lane] == VC2_CBR_MARK;
    for (unsigned long long m = __ballot(marked); m; m &= m - 1) search(base + __ffsll((long long)m) - 1);
  }
}

// The same search with the slice in registers: a lane keeps its eight luma and eight chroma magnitudes as floats for all
// trials, the quantiser is one multiply by a rounded-up 4 / factor, and the code length comes from the exponent of
// 4|c| / factor + 1 -- no LDS staging, no integer multiplies, no count-leading-zeros.
//   quant (Quantisation.cpp:69-76): q = floor(4|c| / f).  With r = the smallest float >= 4 / f and |c| < 2^15:
//     fl(|c| * r) lies in [4|c|/f, 4|c|/f * (1 + 2^-22)); 4|c|/f is a multiple of 1/f, so no integer lies in between as
//     long as 4|c| < 2^22 -> truncation gives q exactly.
//   SignedVLC bits (VLC.cpp:78-85): 1 for q = 0, else 2 floor(log2(q + 1)) + 2, and floor(log2(q + 1)) =
//     floor(log2(x + 1)) for any real x in [q, q + 1): E = exponent of fma(|c|, r, 1), exact while
//     1.5 * 4|c| * 2^-23 + f * 2^-23 < 1 (f < 2^22: quantiser indices up to 79); bits = 2E + 1 + min(E, 1).
// Anything outside that domain (a coefficient beyond 16 bits, a trial index above 79, a length byte overflow) hands the
// slice to the general kernel through VC2_CBR_MARK: same answers, found the slow way.
// Needs the common geometry (components of at most 512 / 256 coefficients, multiples of 8).
#ifndef CBR_SPW
#define CBR_SPW 8 // (measured: 4 -> 0.57, 8 -> 0.54, 16 -> 0.53 ms per 16 UHD pictures; one slice, as before: 0.76) consecutive slices per wavefront (the first is bisected, the others start at their predecessor's threshold)
#endif
template <class ST>
__global__ __launch_bounds__(256) void k_cbr_search_reg(const CbrParams p) {
  __shared__ __attribute__((aligned(8))) unsigned char band_lds[768];
  unsigned char *band_y = band_lds, *band_c = band_lds + 512;
  __shared__ uint4 s_tab[80];  // by quantiser index: (rounded-up 4 / factor as a float, factor, offset + 2, -)
  __shared__ int s_qm[32];     // 16 x the quantisation matrix entry of every subband (byte offsets into s_tab)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slice0 = (blockIdx.x * 4 + wave) * CBR_SPW, pic = blockIdx.y;
  {
    copy_band_lut(band_y, p.band_lut);
    if (threadIdx.x < 80)
      s_tab[threadIdx.x] = make_uint4(__float_as_uint(c_qs.inv4[threadIdx.x]), (unsigned)c_qs.qf[threadIdx.x], (unsigned)c_qs.off[threadIdx.x] + 2u, 0u);
    if (threadIdx.x < 32) s_qm[threadIdx.x] = threadIdx.x < p.n_bands ? 16 * p.qmatrix[threadIdx.x] : 0;
  }
  __syncthreads();
  const int half = lane >> 5, cc = 1 + half; // lanes 0-31 U, lanes 32-63 V
  const int jy = lane * 8, jc = (lane & 31) * 8;
  const bool has_y = jy < p.comp_n[0], has_c = jc < p.comp_n[1];
  int guess = -1; // the previous slice's threshold (see the search below)
  for (int slice = slice0; slice < min(slice0 + CBR_SPW, p.n_slices); ++slice) { // no workgroup barriers below
  const size_t rec_at = (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs;
  const ST *rec = (const ST *)p.store + rec_at;
  const int32_t *recw = St<ST>::narrow ? p.store_wide + rec_at : nullptr;
  float fy[8], fc[8]; // |coefficient|
  int my[8], mc[8];   // 16 x the matrix entry of its subband: table offset of a trial tq = clamp(16 tq - m, 0, 16 * 79)
  bool out = false;
  {
    int raw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = 0;
    if (has_y) St<ST>::load8(rec + p.comp_off[0] + jy, recw + p.comp_off[0] + jy, raw);
    const uint2 b = *(const uint2 *)(band_y + (has_y ? jy : 0));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int a = raw[k] < 0 ? -raw[k] : raw[k];
      out |= (unsigned)a > 32767u;
      fy[k] = (float)a;
      my[k] = s_qm[((k < 4 ? b.x : b.y) >> (8 * (k & 3))) & 0x1Fu];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = 0;
    if (has_c) St<ST>::load8(rec + p.comp_off[cc] + jc, recw + p.comp_off[cc] + jc, raw);
    const uint2 d = *(const uint2 *)(band_c + (has_c ? jc : 0));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int a = raw[k] < 0 ? -raw[k] : raw[k];
      out |= (unsigned)a > 32767u;
      fc[k] = (float)a;
      mc[k] = s_qm[((k < 4 ? d.x : d.y) >> (8 * (k & 3))) & 0x1Fu];
    }
  }
  const int avail = p.slice_bytes[slice] - 4;
  const char *tab = (const char *)s_tab;
  auto entry = [&](int tq16, int m) -> int { return min(max(tq16 - m, 0), 16 * 79); };
  auto bits8 = [&](int tq16, const float (&f)[8], const int (&m)[8], bool has, int &sum, int &last_end) {
    sum = 0; last_end = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float y = __builtin_fmaf(f[k], *(const float *)(tab + entry(tq16, m[k])), 1.0f);
      const int eb = (int)__builtin_amdgcn_ubfe(__float_as_uint(y), 23, 8); // 127 + E
      sum += 2 * eb + min(eb, 128) - 380;                                    // 2E + min(E, 1) + 1
      last_end = eb >= 128 ? sum : last_end;
    }
    if (!has) { sum = 0; last_end = 0; }
  };
  auto comp_bytes = [&](int count, bool &bad) -> int { // the same rounded-up reciprocal for the division by the size scalar
    const int len = (int)((float)(((count + 7) >> 3) + p.scalar - 1) * p.inv_scalar);
    bad |= len > 255;
    return __mul24(len, p.scalar);
  };
  auto need_bytes = [&](int tq, bool &bad) -> int {
    bad |= tq - p.qm_min > 79; // a factor of 2^22 or more: outside the float domain
    int sum, last_end, sum_c, last_c;
    bits8(16 * tq, fy, my, has_y, sum, last_end);
    bits8(16 * tq, fc, mc, has_c, sum_c, last_c);
    // both scans in one: a lane's bits are below 2^9, a component's below 2^15
    const int incl2 = wave_incl_scan(sum | (sum_c << 16), lane);
    const int incl = incl2 & 0xFFFF, incl_c = incl2 >> 16;
    int need = comp_bytes(wave_max(last_end ? incl - sum + last_end : 0), bad);
    const int total_u = __builtin_amdgcn_readlane(incl_c, 31);
    const int rel = incl_c - sum_c - (half ? total_u : 0);
    const int cnt = seg_max<32>(last_c ? rel + last_c : 0);
    need += comp_bytes(__builtin_amdgcn_readlane(cnt, 0), bad) + comp_bytes(__builtin_amdgcn_readlane(cnt, 32), bad);
    return need;
  };
  // luma-only sum of squared reconstruction error (EncodeStream.cpp:73-125 through quant / scale, Quantisation.cpp:69-95)
  auto yss = [&](int tq, bool &bad) -> long long {
    bad |= tq - p.qm_min > 79;
    long long acc = 1ll << 35; // keeps the lane's sum of eight 32-bit products non-negative
#pragma unroll
    for (int g = 0; g < 8; g += 4) { // four table reads in flight
      uint4 t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = *(const uint4 *)(tab + entry(16 * tq, my[g + k]));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned q = (unsigned)(fy[g + k] * __uint_as_float(t[k].x));
        const unsigned r = (__umul24(q, t[k].y) + __umul24(min(q, 1u), t[k].z)) >> 2; // scale(): nothing is added to a zero
        const int d = (int)fy[g + k] - (int)r;
        acc += (long long)__mul24(d, d);
      }
    }
    // the 64-lane sum as two sums of 24-bit limbs on the DPP adder
    const int lo = seg_incl_scan<64>((int)(acc & 0xFFFFFF), lane), hi = seg_incl_scan<64>((int)(acc >> 24), lane);
    return (long long)__builtin_amdgcn_readlane(lo, 63) + ((long long)__builtin_amdgcn_readlane(hi, 63) << 24) - (64ll << 35);
  };

  // The reference bisects: trial 63, steps 32 .. 1, 0 -- seven measurements of the slice (EncodeStream.cpp:88-104).  The
  // bytes a slice needs never grow with the index (every factor grows, so every magnitude, code length and
  // last-non-zero position shrinks or stays), so what the bisection returns is the THRESHOLD T = the smallest index in
  // 0 .. 126 whose bytes fit (127: none), and it raises an error exactly when the smallest trial it visits -- a function
  // of T alone -- overflows a length byte.  Neighbouring slices have neighbouring thresholds: a wavefront takes
  // CBR_SPW consecutive slices, bisects the first like the reference and starts every other one at its predecessor's
  // T: fits(T) and not fits(T - 1) are two measurements instead of seven (further away: gallop, then bisect the bracket),
  // plus one at the reference's smallest trial for the error it would have raised there.
  bool bad = __any(out);
  int trial = 63, q = 127, delta = 64;
  if (guess < 0) {
    while (delta > 0 && !bad) {
      delta >>= 1;
      const int need = need_bytes(trial, bad);
      bad = __any(bad);
      if (need <= avail) { if (trial < q) q = trial; trial -= delta; }
      else trial += delta;
    }
  } else if (!bad) {
    int lo = -1, hi = 127, step = 1, lowest = 127; // lo: the largest index known not to fit; hi: the smallest known to fit
    int t = min(guess, 126);
    for (;;) {
      const int need = need_bytes(t, bad);
      bad = __any(bad);
      if (bad) break;
      lowest = min(lowest, t);
      if (need <= avail) hi = t; else lo = t;
      if (hi - lo <= 1) break;
      if (hi == 127) { if (lo >= 126) break; t = min(126, lo + step); step *= 2; }   // nothing fits yet: up
      else if (lo < 0) { if (hi <= 0) break; t = max(0, hi - step); step *= 2; }      // everything fits so far: down
      else t = (lo + hi) >> 1;
    }
    q = hi;
    if (!bad) { // the smallest trial of the reference's walk to this threshold
      int rt = 63, rd = 64, rmin = 127, rmax = 0;
      while (rd > 0) { rd >>= 1; rmin = min(rmin, rt); rmax = max(rmax, rt); if (rt >= q) rt -= rd; else rt += rd; }
      // (its largest trial may leave this kernel's domain, or the quantiser table -- the error the reference raises there:
      // the general kernel walks the reference's own path for such slices, as before)
      if (rmax - p.qm_min > 79) bad = true;
      else if (rmin < lowest) { (void)need_bytes(rmin, bad); bad = __any(bad); }
    }
  }
  const int q_fit = q; // the threshold (before the refinement below): the next slice's starting point
  if (!bad) {
    trial = q;
    long long prev = yss(trial, bad), d;
    do {
      ++trial;
      const long long cur = yss(trial, bad);
      if (bad) break;
      d = cur - prev;
      prev = cur;
    } while (d < 0);
    q = trial - 1;
  }
  if (lane == 0) p.qidx[(size_t)pic * p.n_slices + slice] = bad ? VC2_CBR_MARK : q;
  guess = bad ? -1 : q_fit;
  } // slices of the wavefront
}

#include "vc2hip_cbr16.h"

// wavefronts per workgroup so that their LDS (per_wave bytes each) fits: 4 down to 1; 0 if even one does not
int vc2_waves_for_lds(size_t per_wave);
int vc2_waves_for_lds(size_t per_wave) {
  if (per_wave == 0) return 4;
  const size_t w = (size_t)(160 * 1024) / per_wave;
  return (int)std::min<size_t>(4, w);
}
void vc2_launch_cbr(Launcher &L, const CbrParams &p0, int n_pictures, hipStream_t s) {
  CbrParams p = p0;
  fill_band_lut(p.band_lut, p.comp_n, p.comp_n0);
  const size_t per_wave = (size_t)p.slice_coefs * 4 + 32 * 16, tables = 768; // + the wavefront's quantiser table; + band tables
  vc2_prof_begin(L, "cbr_search", s);
  p.only_marked = 0;
  p.qm_min = p.qmatrix[0];
  for (int b = 1; b < p.n_bands; ++b) p.qm_min = std::min(p.qm_min, p.qmatrix[b]);
  p.inv_scalar = 1.0f / (float)p.scalar; // the smallest float >= 1 / scalar
  if ((double)p.inv_scalar < 1.0 / (double)p.scalar) p.inv_scalar = nextafterf(p.inv_scalar, INFINITY);
  // (general_only = VC2HIP_FLAG_CBR_GENERAL: A/B and test switch, the general kernel only)
  const bool reg = !p.general_only && p.comp_n[0] <= 512 && p.comp_n[1] <= 256 && p.comp_n[1] == p.comp_n[2] && p.n_bands <= 32 &&
                   p.comp_n[0] % 8 == 0 && p.comp_n[1] % 8 == 0 && (p.store_stride % 8) == 0 && (p.slice_coefs % 8) == 0 &&
                   p.comp_off[1] % 8 == 0 && p.comp_off[2] % 8 == 0;
  if (reg) { // the register kernel, then the general one over the slices it handed back (usually none: a small grid)
    const int per_wg = 4 * CBR_SPW; // slices per workgroup
    static const int use16 = vc2_tune_int("VC2HIP_CBR16", 1);
    CbrParams p16 = p;
    if (use16 && cbr16_plan(p16, p16.lane8)) // the head / run layout of the slice coder: three constants per lane and trial
      VC2_LAUNCH(L, k_cbr_search16, dim3((p.n_slices + per_wg - 1) / per_wg, n_pictures), dim3(256), 0, s, p16);
    else if (p.store16) VC2_LAUNCH(L, k_cbr_search_reg<int16_t>, dim3((p.n_slices + per_wg - 1) / per_wg, n_pictures), dim3(256), 0, s, p);
    else VC2_LAUNCH(L, k_cbr_search_reg<int32_t>, dim3((p.n_slices + per_wg - 1) / per_wg, n_pictures), dim3(256), 0, s, p);
    p.only_marked = 1;
  }
  if (per_wave + tables > 160 * 1024) { // the slice does not fit in LDS: the search reads it from the store in every trial
    const int wpw = 4;
    const size_t lds = tables + wpw * 32 * 16;
    const int gx = p.only_marked ? std::min((p.n_slices + wpw - 1) / wpw, 64) : (p.n_slices + wpw - 1) / wpw;
    if (p.store16) VC2_LAUNCH(L, (k_cbr_search<int16_t, true>), dim3(gx, n_pictures), dim3(64 * wpw), lds, s, p);
    else VC2_LAUNCH(L, (k_cbr_search<int32_t, true>), dim3(gx, n_pictures), dim3(64 * wpw), lds, s, p);
    vc2_prof_end(L, s);
    return;
  }
  const int wpw = std::max(1, std::min(4, (int)((160 * 1024 - tables) / per_wave)));
  const int gx = p.only_marked ? std::min((p.n_slices + wpw - 1) / wpw, 64) : (p.n_slices + wpw - 1) / wpw;
  if (p.store16) {
    vc2_allow_lds((const void *)k_cbr_search<int16_t>, 160 * 1024);
    VC2_LAUNCH(L, k_cbr_search<int16_t>, dim3(gx, n_pictures), dim3(64 * wpw), wpw * per_wave + tables, s, p);
  } else {
    vc2_allow_lds((const void *)k_cbr_search<int32_t>, 160 * 1024);
    VC2_LAUNCH(L, k_cbr_search<int32_t>, dim3(gx, n_pictures), dim3(64 * wpw), wpw * per_wave + tables, s, p);
  }
  vc2_prof_end(L, s);
}

// ------------------------------------------------------------------------------------------
// HQ unpack: one lane per slice component
// ------------------------------------------------------------------------------------------
// Bounded bit reader over 64-bit big-endian words with one word of lookahead: `acc` holds `have`
// unread bits (top aligned), `nxt` the following 64 bits, so peek() always shows 64 valid bits and
// a whole token (zero run + one code, <= 48 bits) is consumed with a single refill check.
// Bytes past the component bound read as 0xFF (VLC.cpp:182-185).
struct WordReader {
  const uint2 *pw;     // next word to fetch (8-byte aligned)
  int left;            // data BITS from the start of that word to the end of the bounded data (<= 0: past the end)
  unsigned long long acc, nxt;
  int have;
  // (16 bytes per load, to visit a payload line 8 times instead of 16, measured 3 % slower: 0.495 against 0.478 ms)
  __device__ __forceinline__ unsigned long long fetch() {
    unsigned long long v = ~0ull; // bits past the bound read as 1 (VLC.cpp:182-185)
    if (left > 0) {
      const uint2 w = *pw;
      v = ((unsigned long long)__builtin_bswap32(w.x) << 32) | __builtin_bswap32(w.y);
      if (left < 64) v |= ~0ull >> left;
    }
    ++pw;
    left -= 64;
    return v;
  }
  // nbits (> 0) of data starting bitoff (0..7) bits into *data
  __device__ __forceinline__ void init_bits(const uint8_t *data, int bitoff, int nbits) {
    const size_t a = (size_t)data;
    const int lead = 8 * (int)(a & 7) + bitoff; // 0..63: word 0 starts `lead` bits before the data
    pw = (const uint2 *)(a & ~(size_t)7);
    left = nbits + lead;
    acc = fetch() << lead;
    have = 64 - lead;
    nxt = fetch();
  }
  __device__ __forceinline__ void init(const uint8_t *data, int nbytes) { init_bits(data, 0, 8 * nbytes); }
  __device__ __forceinline__ void init_ones() { pw = nullptr; left = 0; acc = nxt = ~0ull; have = 64; }
  // the next 64 unread bits; have is 1..64, so both shifts are in range
  __device__ __forceinline__ unsigned long long peek() const { return acc | ((nxt >> 1) >> (have - 1)); }
  __device__ __forceinline__ void skip(int n) { // n <= 64
    if (n < have) { acc <<= n; have -= n; }
    else {
      const int r = n - have; // 0..63 bits into nxt
      acc = nxt << r;
      have = 64 - r;
      nxt = fetch();
    }
  }
};

// HQ unpack v2: one lane per slice component, grouped by component type so that all lanes of a
// wavefront decode the same number of coefficients.  Decoded values are staged 16 per lane in LDS
// and flushed with 16-byte stores (four lanes cover one component's 64-byte run).
// gather the bits at even positions (0,2,4,...) of a 32-bit word into the low half
__device__ __forceinline__ unsigned compact_even32(unsigned x) {
  x &= 0x55555555u;
  x = (x | (x >> 1)) & 0x33333333u;
  x = (x | (x >> 2)) & 0x0F0F0F0Fu;
  x = (x | (x >> 4)) & 0x00FF00FFu;
  x = (x | (x >> 8)) & 0x0000FFFFu;
  return x;
}

// codes of up to 10 bits (|value| <= 30) by their leading 10 bits: length << 8 | value (8-bit two's complement); 0 = longer
__device__ __forceinline__ void vlut_init(unsigned short *vlut) {
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
    unsigned e = 0;
    for (int K = 1; K <= 4 && !e; ++K) {
      bool ok = ((i >> (9 - 2 * K)) & 1) == 1; // terminator after K pairs ...
      for (int j = 0; j < K; ++j) ok = ok && ((i >> (9 - 2 * j)) & 1) == 0; // ... whose flag bits are all 0
      if (ok) {
        unsigned m = 1;
        for (int j = 0; j < K; ++j) m = (m << 1) | ((i >> (8 - 2 * j)) & 1);
        m -= 1;
        const int v = ((i >> (8 - 2 * K)) & 1) ? -(int)m : (int)m;
        e = ((unsigned)(2 * K + 2) << 8) | ((unsigned)v & 0xFFu);
      }
    }
    vlut[i] = (unsigned short)e;
  }
}

// One round of the exp-Golomb decoder: `room` (<= UNP_N) coefficients from br into the lane's staging row st.
// T = int: the values as they are.  T = short (16-bit store): a value outside 16 bits leaves the sentinel in the row and
// goes to wide[position in the round] (vc2hip_store.h); only codes beyond the 10-bit table can be that large.
template <int UNP_N, class T>
__device__ __forceinline__ void decode_round(WordReader &br, int room, T *st, const unsigned short *vlut, int32_t *wide = nullptr) {
  // zero coefficients are the common case: clear the row, then only non-zero values are stored
#pragma unroll
  for (int k = 0; k < UNP_N; k += 16 / (int)sizeof(T)) *(int4 *)(st + k) = make_int4(0, 0, 0, 0);
  auto put = [&](int at, int v) {
    if constexpr (sizeof(T) == 2) {
      if (!St<int16_t>::fits(v)) { wide[at] = v; v = VC2_ST_SENTINEL; }
    }
    st[at] = (T)v;
  };
  int cnt = 0;
  while (cnt < room) {
    const unsigned long long win = br.peek();
    // run of '1' bits = run of zero coefficients (VLC.cpp:283-295: a lone '1' is the value 0), at most 32 per turn so
    // that the code behind it lies inside the window
    const int lz = __clzll((long long)~win);
    const int z = min(min(lz, room - cnt), 32);
    cnt += z;
    // a non-zero coefficient follows unless the round is full or the run goes on; one code path for all cases
    const bool nz = cnt < room && z == lz;
    // non-zero: (0 b)^K 1 s ; follow bits sit at even offsets from the code start
    const unsigned hi = (unsigned)((win << z) >> 32);
    const unsigned follow = hi & 0xAAAAAAAAu;
    if (nz && follow == 0) { // code longer than 32 bits (outside the reference's domain): bit-serial, wraps like the oracle
      br.skip(z);
      unsigned value = 1;
      for (;;) {
        const int f = (int)(br.peek() >> 63);
        br.skip(1);
        if (f) break;
        value = (value << 1) | (unsigned)(br.peek() >> 63);
        br.skip(1);
      }
      value -= 1u;
      int r = 0;
      if (value) {
        r = (br.peek() >> 63) ? (int)(0u - value) : (int)value;
        br.skip(1);
      }
      put(cnt++, r);
      continue;
    }
    int val, len;
    bool big1 = false; // beyond the table: may need the escape
    const unsigned e1 = vlut[hi >> 22];
    if (e1) { val = __builtin_amdgcn_sbfe((int)e1, 0, 8); len = (int)(e1 >> 8); }
    else {
      big1 = true;
      const int K = __clz((int)(follow | 1u)) >> 1;              // 1..15 for a real code (bit 31 of hi is 0 there)
      const unsigned body = hi >> ((32 - 2 * K) & 31);           // top 2K bits: (0 b) pairs
      const unsigned mag = ((1u << K) | compact_even32(body)) - 1u;
      const int neg = (int)((hi >> ((30 - 2 * K) & 31)) & 1u);
      val = neg ? (int)(0u - mag) : (int)mag;
      len = 2 * K + 2;
    }
    if (nz) {
      if (sizeof(T) == 2 && big1) put(cnt, val);
      else st[cnt] = (T)val;
    }
    cnt += nz ? 1 : 0;
    int n = nz ? z + len : z;
    // a second token from the same window when it lies wholly inside it (saves a window build, a refill test and a
    // loop turn per pair): zero run, then a code with its terminator and sign inside the 32 bits examined
    if (n <= 31) {
      const unsigned long long w2 = win << n;
      const int z2 = min(__clzll((long long)~w2), room - cnt);
      const unsigned hi2 = (unsigned)((w2 << z2) >> 32);
      const unsigned follow2 = hi2 & 0xAAAAAAAAu;
      // n + z2 + 32 <= 64 keeps every examined bit a real stream bit
      const bool take = nz && n + z2 <= 32 && cnt + z2 < room && follow2 != 0;
      if (take) {
        int val2, len2;
        bool big2 = false;
        const unsigned e2 = vlut[hi2 >> 22];
        if (e2) { val2 = __builtin_amdgcn_sbfe((int)e2, 0, 8); len2 = (int)(e2 >> 8); }
        else {
          big2 = true;
          const int K2 = __clz((int)follow2) >> 1;
          const unsigned body2 = hi2 >> ((32 - 2 * K2) & 31);
          const unsigned mag2 = ((1u << K2) | compact_even32(body2)) - 1u;
          const int neg2 = (int)((hi2 >> ((30 - 2 * K2) & 31)) & 1u);
          val2 = neg2 ? (int)(0u - mag2) : (int)mag2;
          len2 = 2 * K2 + 2;
        }
        cnt += z2;
        if (sizeof(T) == 2 && big2) put(cnt, val2);
        else st[cnt] = (T)val2;
        ++cnt;
        n += z2 + len2;
      }
    }
    br.skip(n);
  }
}

// UNP_N coefficients are staged per lane and round: 16 = 64-byte runs, 32 = whole 128-byte lines (no
// read-for-ownership of the other half line, but twice the LDS: 4 instead of 6 wavefronts per SIMD).
template <int UNP_N, bool NT = true>
__global__ __launch_bounds__(256) void k_hq_unpack(const UnpackParams p) {
  constexpr int UNP_PITCH = UNP_N + 4; // ints per staging row, 16-byte aligned rows
  __shared__ __attribute__((aligned(16))) int stage[4][64 * UNP_PITCH];
  __shared__ unsigned long long outp[4][64];
  // codes of up to 10 bits (|value| <= 30) by their leading 10 bits: length << 8 | value (8-bit two's complement); 0 = longer
  __shared__ unsigned short vlut[1024];
  vlut_init(vlut);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pic = blockIdx.y, comp = blockIdx.z;
  const int slice = blockIdx.x * 256 + threadIdx.x;
  const bool active = slice < p.n_slices;
  const int n = p.comp_n[comp];
  int *st = stage[wave] + lane * UNP_PITCH;
  WordReader br;
  {
    int32_t *out = nullptr;
    const uint8_t *data = nullptr;
    unsigned len = 0;
    if (active) {
      const uint8_t *pay = p.payload + (size_t)pic * p.payload_stride;
      const unsigned long long plen = min(p.lens[pic], (unsigned long long)p.payload_stride); // never past the picture's slot
      unsigned long long pos = (unsigned long long)p.offsets[(size_t)pic * p.n_slices + slice] + p.prefix;
      auto rd = [&](unsigned long long a) -> unsigned { return a < plen ? pay[a] : 0u; };
      if (comp == 0) p.qidx[(size_t)pic * p.n_slices + slice] = (int)rd(pos);
      pos += 1;
      for (int c = 0; c < comp; ++c) pos += 1 + (unsigned long long)rd(pos) * p.scalar;
      len = rd(pos) * p.scalar;
      pos += 1;
      if (pos + len > plen) {
        atomicOr(p.err, VC2_DEVERR_STREAM);
        len = pos < plen ? (unsigned)(plen - pos) : 0;
      }
      data = pay + (pos < plen ? pos : 0);
      out = (int32_t *)p.store + (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs + p.comp_off[comp];
    }
    outp[wave][lane] = (unsigned long long)out;
    if (active) br.init(data, (int)len); else br.init_ones();
  }
  for (int base = 0; base < n; base += UNP_N) {
    const int room = min(UNP_N, n - base);
    decode_round<UNP_N, int>(br, room, st, vlut);
    // flush: 4 lanes x 16 bytes per component run.  The staging rows are private to the wavefront, so only
    // its own lanes have to agree (LDS operations of one wavefront execute in order): no workgroup barrier,
    // the four wavefronts of the workgroup drift apart freely.
    wave_lds_sync();
    const int *sw = stage[wave];
    constexpr int LPR = UNP_N / 4; // lanes per component run
#pragma unroll
    for (int j = 0; j < LPR; ++j) {
      const int r = j * (64 / LPR) + lane / LPR, c = (lane % LPR) * 4;
      int32_t *dst = (int32_t *)outp[wave][r];
      if (dst && c < room) {
        const int4 v = *(const int4 *)(sw + r * UNP_PITCH + c);
        int *d = dst + base + c;
        if constexpr (NT) {
          typedef int v4i __attribute__((ext_vector_type(4)));
          const v4i vv = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(vv, (__attribute__((address_space(1))) v4i *)(size_t)d); // one 16-byte streaming global store per lane
        } else *(int4 *)d = v;
      }
    }
    wave_lds_sync();
  }
}

// ------------------------------------------------------------------------------------------
// HQ unpack into the 16-bit store: table-driven, several coefficients per look-up.
//
// A lane decodes one slice component as before, but a turn of its loop is three look-ups of the next 10 stream bits in a
// table of "what these bits decode to": bits consumed, coefficients produced (zeros included, at most 8), and up to
// two non-zero values with their positions -- a few coefficients per look-up for ~20 instructions, where the
// code-by-code decoder above spends ~150 per turn of two codes.  What the table cannot hold (a code longer than 10 bits:
// |value| > 30) takes a separate step after the look-ups.  (While three wavefronts shared a SIMD that step ran on every
// fourth turn only -- with 64 lanes per wavefront something rare per lane happens on almost every turn somewhere, and a
// step the wavefront executes for one lane costs all of them; at six wavefronts per SIMD a stalled lane's waiting costs
// more than the step: every turn, 16 UHD pictures 0.300 -> 0.295 ms, 32 HD pictures 0.225 -> 0.190.  UNP_LONG_EVERY = 2^k - 1
// brings the cadence back.)
// The bit reader keeps 33..64 unread bits in a register pair and appends one pre-fetched 32-bit word when it runs low
// (one short conditional block per turn instead of the 64-bit window assembly and refill of WordReader).
// Rows hold 32 coefficients plus 8 of slack: a table entry is applied whole, coefficients that spill over the end of a
// round are carried into the next one.  (Rows of 64 make every flush a whole 128-byte line of the store, but the rows of
// a workgroup then take 37 KB of LDS and three wavefronts per SIMD is all a CU holds; with 32 it holds six.  16 UHD-1 / 32
// HD / 4 UHD-2 pictures: 0.310 / 0.285 / 0.39 ms with rows of 64, 0.302 / 0.226 / 0.343 ms with rows of 32 and the
// second request in flight below; rows of 16: 0.32 / 0.215 / 0.37.)
// ------------------------------------------------------------------------------------------
// entry: bits [3:0] consumed (1..10), [7:4] coefficients (1..8), [11:8] / [15:12] positions of the two values,
// [23:16] / [31:24] the values (int8).  No non-zero value: both slots store 0 at position 0 (a zero anyway); one: both
// slots hold it.  0: the first token is a code longer than the index.
#ifndef UNP_LOOKS
#define UNP_LOOKS 4 // look-ups per turn of k_hq_unpack16 (2: 0.570, 3: 0.545, 4: 0.535, 5: 0.538, 6: 0.550 ms per 32 UHD pictures; the fourth only with a whole index of unread bits left)
#endif
#ifndef VC2_UNP16_N
#define VC2_UNP16_N 32
#endif
#ifndef UNP_LONG_EVERY
#define UNP_LONG_EVERY 0
#endif
#ifndef VC2_UNP_LUT_BITS
#define VC2_UNP_LUT_BITS 10
#endif
constexpr int UNP_LUT_BITS = VC2_UNP_LUT_BITS, UNP_LUT_N = 1 << UNP_LUT_BITS;
__device__ unsigned g_unp_lut[UNP_LUT_N];
void vc2_upload_unpack_lut(hipStream_t s) {
  // built once (contexts are created from several worker threads at a time: the copy below may still be reading it)
  static unsigned host[UNP_LUT_N];
  static std::once_flag once;
  std::call_once(once, [] {
  for (int idx = 0; idx < UNP_LUT_N; ++idx) {
    auto bit = [&](int i) { return (idx >> (UNP_LUT_BITS - 1 - i)) & 1; };
    int pos = 0, nc = 0, nnz = 0, at[2] = {0, 0}, val[2] = {0, 0};
    while (nc < 8 && pos < UNP_LUT_BITS) {
      if (bit(pos)) { ++nc; ++pos; continue; } // a lone '1': the value 0 (VLC.cpp:283-295)
      int q = pos, mag = 1;
      bool whole = false;
      while (q + 1 < UNP_LUT_BITS) {           // (0 b)* 1 s
        mag = (mag << 1) | bit(q + 1);
        q += 2;
        if (q >= UNP_LUT_BITS) break;
        if (bit(q)) { whole = q + 1 < UNP_LUT_BITS; break; }
      }
      if (!whole || nnz == 2) break;
      const int v = bit(q + 1) ? -(mag - 1) : mag - 1;
      at[nnz] = nc; val[nnz] = v; ++nnz;
      ++nc;
      pos = q + 2;
    }
    if (nnz == 1) { at[1] = at[0]; val[1] = val[0]; }
    host[idx] = nc == 0 ? 0u
                        : (unsigned)pos | (unsigned)nc << 4 | (unsigned)at[0] << 8 | (unsigned)at[1] << 12 |
                              (unsigned)(val[0] & 0xFF) << 16 | (unsigned)(val[1] & 0xFF) << 24;
  }
  });
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_unp_lut), host, sizeof host, 0, hipMemcpyHostToDevice, s);
}

// 33..64 unread bits at the top of `acc`; behind them a queue of up to four stream words in registers, and behind that
// the next sixteen bytes of the stream, already requested.  A lane reads ITS OWN stream: the 64 loads of a wavefront
// instruction go to 64 different cache lines, and with one dword per load every 128-byte line of the payload crossed
// the L2 -> L1 path 32 times (the streams of a CU's wavefronts do not fit its 32 KiB L1: 3.4 - 6.7 x the payload even
// from beyond L2, round 2).  Sixteen bytes per load make that a quarter of the requests, each a quarter as often, and
// put two to five words of distance between a request and its use.  Words are dword-aligned (a wave-uniform base + a
// 32-bit offset per lane); bits past the bounded data read as 1 (VLC.cpp:182-185) and no word without a data bit is
// ever loaded.
#ifndef VC2_UNP_DEEP
#define VC2_UNP_DEEP 1
#endif
#ifndef VC2_UNP_TAIL16
#define VC2_UNP_TAIL16 1
#endif
#ifndef VC2_UNP_SELECT
// 0: the round-4 refill.  1: the requested words selected in place instead of shifted down a piece per refill.  2: and the
// request and its wait written as asm (Reader32::skip's first form), which does move the wait from two instructions behind
// the request to the next refill of any lane -- and is SLOWER: 32 UHD pictures 0.446 -> 0.466 ms, 4 UHD-2 0.253 -> 0.267,
// 32 HD 0.155 -> 0.166 (same box, alternating; 1: 0.434 -> 0.492, ten more registers and a wavefront less per SIMD).  At six
// wavefronts per SIMD the others fill a wavefront's wait; the twelve selections and two wave-uniform tests per refill cost
// more than it.  Kept as the record of VERDICT round 4, item 4; bit-exact (tests/test_gpu_parity.py, test_gpu_wide.py).
#define VC2_UNP_SELECT 0
#endif
#ifndef VC2_UNP_PAIR
#define VC2_UNP_PAIR 2 // the stream's requests FOUR at a time (64 adjacent bytes) on every fourth refill; 1: two on every second; 0: one per refill (see Reader32::skip)
#endif
struct Reader32 {
  unsigned long long acc;
  int have;                // valid bits in acc, 33..64 between turns
  unsigned q0, q1, q2, q3; // the words after those in acc (q0 first); qn of them are valid, 1..4 between turns
  int qn;
  unsigned n0, n1, n2, n3; // the four words after the queue (these and the ones below: in memory order, byte-swapped on their way into the queue)
#if VC2_UNP_DEEP
  unsigned m0, m1, m2, m3; // and the four after those ...
#if VC2_UNP_PAIR
  unsigned o0, o1, o2, o3; // ... and four more: requests go out TWO AT A TIME, for 32 adjacent bytes (round 4, see skip())
  int pairs;               // parity of the refills: the odd ones only shift (VC2_UNP_PAIR == 2: their count mod 4)
#if VC2_UNP_PAIR == 2
  unsigned r0, r1, r2, r3, s0, s1, s2, s3; // FOUR requests (64 adjacent bytes) on every fourth refill
#if VC2_UNP_SELECT == 2
  typedef unsigned u4_t __attribute__((ext_vector_type(4)));
  u4_t M, O, R, S; // (the same sixteen words as whole pieces: the destinations of the requests written in skip(); m0 .. s3 unused)
#endif
#endif
#endif
#endif
  unsigned off;            // byte offset (from the payload base) of the word after those
  int left;                // data bits from that word on (<= 0: none)
  unsigned safe;           // end of the picture's payload slot, as such an offset (wave-uniform)
#ifdef VC2HIP_ABLATE
  unsigned dbg_mask;
#endif
  __device__ __forceinline__ unsigned fetch1(const uint8_t *pay) {
    unsigned v = ~0u;
    if (left > 0) {
      v = __builtin_bswap32(*(const unsigned *)(pay + off));
      if (left < 32) v |= ~0u >> left;
    }
    off += 4;
    left -= 32;
    return v;
  }
  __device__ __forceinline__ void fetch4(const uint8_t *pay, unsigned &a, unsigned &b, unsigned &c, unsigned &d) {
    if (left >= 128) { // sixteen bytes of data: one load
#ifdef VC2HIP_ABLATE
      const Dword4 v = *(const Dword4 *)(pay + (off & dbg_mask));
#else
      const Dword4 v = *(const Dword4 *)(pay + off);
#endif
      // (the words stay in memory order until they move into the queue -- skip() / init() swap them there: swapped here,
      // four more instructions stand between the request and the wait the compiler puts behind it, in every refill)
      a = v.x; b = v.y; c = v.z; d = v.w;
      off += 16;
      left -= 128;
    }
#if VC2_UNP_TAIL16
    // The stream's last data bytes: still ONE load when its sixteen bytes lie inside the picture's slot (`safe`), the bits
    // behind the data forced to 1.  Word by word (below) every load was waited for before the next went out, and a
    // wavefront meets the end of a stream once per lane: 64 times up to four trips to L2 in a row.
    else if (left > 0 && off + 16u <= safe) {
      const Dword4 v = *(const Dword4 *)(pay + off);
      auto ones_from = [](int k) -> unsigned { return k >= 32 ? 0u : ~0u >> max(k, 0); }; // bits k.. of a word (k = its data bits)
      a = v.x | __builtin_bswap32(ones_from(left));
      b = v.y | __builtin_bswap32(ones_from(left - 32));
      c = v.z | __builtin_bswap32(ones_from(left - 64));
      d = v.w | __builtin_bswap32(ones_from(left - 96));
      off += 16;
      left -= 128;
    }
#endif
    else { // the slot's end (or no data left): word by word, back into memory order like the others
      a = __builtin_bswap32(fetch1(pay)); b = __builtin_bswap32(fetch1(pay)); c = __builtin_bswap32(fetch1(pay)); d = __builtin_bswap32(fetch1(pay));
    }
  }
  __device__ __forceinline__ void fetch4(const uint8_t *pay) { fetch4(pay, n0, n1, n2, n3); }
  // nbytes of data at byte offset pos of the payload
  __device__ __forceinline__ void init(const uint8_t *pay, unsigned pos, int nbytes) {
    const int lead = 8 * (int)(pos & 3u);
    off = pos & ~3u;
    left = nbytes > 0 ? 8 * nbytes + lead : 0;
    fetch4(pay);
    acc = (((unsigned long long)__builtin_bswap32(n0) << 32) | __builtin_bswap32(n1)) << lead;
    have = 64 - lead;
    q0 = __builtin_bswap32(n2); q1 = __builtin_bswap32(n3); q2 = ~0u; q3 = ~0u; qn = 2;
    fetch4(pay);
#if VC2_UNP_DEEP
    fetch4(pay, m0, m1, m2, m3);
#if VC2_UNP_PAIR
    fetch4(pay, o0, o1, o2, o3);
    pairs = 0;
#if VC2_UNP_PAIR == 2
    fetch4(pay, r0, r1, r2, r3);
    fetch4(pay, s0, s1, s2, s3);
#if VC2_UNP_SELECT == 2
    M = u4_t{m0, m1, m2, m3}; O = u4_t{o0, o1, o2, o3}; R = u4_t{r0, r1, r2, r3}; S = u4_t{s0, s1, s2, s3};
#endif
#endif
#endif
#endif
  }
  __device__ __forceinline__ unsigned top() const { return (unsigned)(acc >> 32); }
#if VC2_UNP_SELECT == 2 && VC2_UNP_PAIR == 2 && VC2_UNP_DEEP
  // Round 5: request and wait written as asm, at the top level of the turn (the round-4 form below says what the queue
  // is).  Left to the compiler, the loads of a request go to temporary registers that are copied into the queue's behind
  // s_waitcnt vmcnt(0), two instructions behind the request: the wavefront stood still for the whole trip to memory whenever
  // one of its lanes requested.  Here the sixteen requested words stay where the loads put them (four pieces M, O, R, S; the
  // next four words are SELECTED by the count of refills instead of shifted down), the asm statements take the pieces as
  // read-write operands -- the compiler knows nothing of a pending load and has no copy to make -- and the wait stands in
  // front of the only reader, the selection at any lane's next refill: about a turn behind the request.  The requesting
  // lanes are an EXEC mask inside the statement; the statements themselves stand under wave-uniform tests only.
  // (The compiler's own counted waits stay right: the counter is in order, and its count of the operations behind one of
  // its own loads can only be too small with these four among them -- it waits longer, never shorter.)
  __device__ __forceinline__ void skip(const uint8_t *pay, int n) { // n <= 32
    acc <<= n;
    have -= n;
    bool refill = false;
    if (have <= 32) {
      acc |= (unsigned long long)q0 << (32 - have);
      have += 32;
      q0 = q1; q1 = q2; q2 = q3;
      refill = --qn == 0;
    }
    if (__builtin_amdgcn_ballot_w64(refill) == 0ull) return; // (wave-uniform)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(M), "+v"(O), "+v"(R), "+v"(S));
    bool request = false;
    if (refill) {
      q0 = __builtin_bswap32(n0); q1 = __builtin_bswap32(n1); q2 = __builtin_bswap32(n2); q3 = __builtin_bswap32(n3); qn = 4;
      // (two bit tests per word: a chain of comparisons with 0, 1, 2 becomes a table in scratch memory)
      const bool b0 = pairs & 1, b1 = pairs & 2;
      auto pick = [&](unsigned m, unsigned o, unsigned r, unsigned s) -> unsigned { const unsigned lo = b0 ? o : m, hi = b0 ? s : r; return b1 ? hi : lo; };
      n0 = pick(M.x, O.x, R.x, S.x); n1 = pick(M.y, O.y, R.y, S.y); n2 = pick(M.z, O.z, R.z, S.z); n3 = pick(M.w, O.w, R.w, S.w);
      request = ((++pairs) & 3) == 0;
    }
    const bool whole = request && left >= 512; // four whole pieces: one test for the four requests
    const unsigned long long wmask = __builtin_amdgcn_ballot_w64(whole);
    if (wmask) { // (wave-uniform)
#ifdef VC2HIP_ABLATE
      const unsigned o_ = off & dbg_mask;
#else
      const unsigned o_ = off;
#endif
      unsigned long long saved;
      asm volatile("s_and_saveexec_b64 %4, %7\n\t"
                   "global_load_dwordx4 %0, %5, %6\n\tglobal_load_dwordx4 %1, %5, %6 offset:16\n\t"
                   "global_load_dwordx4 %2, %5, %6 offset:32\n\tglobal_load_dwordx4 %3, %5, %6 offset:48\n\t"
                   "s_mov_b64 exec, %4"
                   : "+v"(M), "+v"(O), "+v"(R), "+v"(S), "=&s"(saved) : "v"(o_), "s"(pay), "s"(wmask) : "scc");
    }
    if (whole) { off += 64; left -= 512; }
    else if (request) { // the stream's end: piece by piece (fetch4's other forms)
      unsigned a, b, c, d;
      fetch4(pay, a, b, c, d); M = u4_t{a, b, c, d};
      fetch4(pay, a, b, c, d); O = u4_t{a, b, c, d};
      fetch4(pay, a, b, c, d); R = u4_t{a, b, c, d};
      fetch4(pay, a, b, c, d); S = u4_t{a, b, c, d};
    }
  }
#else
  __device__ __forceinline__ void skip(const uint8_t *pay, int n) { // n <= 32
    acc <<= n;
    have -= n;
    if (have <= 32) {
      acc |= (unsigned long long)q0 << (32 - have);
      have += 32;
      q0 = q1; q1 = q2; q2 = q3;
#if VC2_UNP_DEEP
      // A lane's stream advances 16 bytes per refill, and between two refills the other 1500 lanes of the CU push its
      // 128-byte line out of L1 and mostly out of L2: with one 16-byte request per refill a payload line was fetched up
      // to eight times (rocprofv3 FETCH_SIZE 3.2 - 3.9 x the payload).  Every second refill now requests the next 32
      // bytes at once (two adjacent loads, the second meets the first one's line), the others only shift: half the
      // requests per line, the same eight words or more between a request and its use.
      // Round 4's end: what a request costs is not its traffic but the WAIT the compiler puts directly behind it (the
      // results of the forms of request merge through temporary registers that are copied out at once): whenever any lane of a
      // wavefront requests, the wavefront stands still for that trip to memory and for the stores of its last flush (the
      // counter is in order).  So the fewer refills request at all, the better: one request per refill 0.536 ms per 32 UHD
      // pictures, two on every second 0.468, four on every fourth 0.469 (cfg 4, whose streams are long: 0.326 / 0.265 / 0.243).
#if VC2_UNP_PAIR == 2
      if (--qn == 0) {
        q0 = __builtin_bswap32(n0); q1 = __builtin_bswap32(n1); q2 = __builtin_bswap32(n2); q3 = __builtin_bswap32(n3); qn = 4;
#if VC2_UNP_SELECT
        {
          // (two bit tests per word: a chain of comparisons with 0, 1, 2 becomes a table in scratch memory)
          const bool b0 = pairs & 1, b1 = pairs & 2;
          auto pick = [&](unsigned m, unsigned o, unsigned r, unsigned s) -> unsigned { const unsigned lo = b0 ? o : m, hi = b0 ? s : r; return b1 ? hi : lo; };
          n0 = pick(m0, o0, r0, s0); n1 = pick(m1, o1, r1, s1); n2 = pick(m2, o2, r2, s2); n3 = pick(m3, o3, r3, s3);
        }
#else
        n0 = m0; n1 = m1; n2 = m2; n3 = m3;
        m0 = o0; m1 = o1; m2 = o2; m3 = o3;
        o0 = r0; o1 = r1; o2 = r2; o3 = r3;
        r0 = s0; r1 = s1; r2 = s2; r3 = s3;
#endif
        if (((++pairs) & 3) == 0) {
          if (left >= 512) { // four whole pieces: one test for the four requests
#ifdef VC2HIP_ABLATE
            const uint8_t *at = pay + (off & dbg_mask);
#else
            const uint8_t *at = pay + off;
#endif
            const Dword4 a = *(const Dword4 *)at, b = *(const Dword4 *)(at + 16), c = *(const Dword4 *)(at + 32), d = *(const Dword4 *)(at + 48);
            m0 = a.x; m1 = a.y; m2 = a.z; m3 = a.w; o0 = b.x; o1 = b.y; o2 = b.z; o3 = b.w;
            r0 = c.x; r1 = c.y; r2 = c.z; r3 = c.w; s0 = d.x; s1 = d.y; s2 = d.z; s3 = d.w;
            off += 64;
            left -= 512;
          } else { fetch4(pay, m0, m1, m2, m3); fetch4(pay, o0, o1, o2, o3); fetch4(pay, r0, r1, r2, r3); fetch4(pay, s0, s1, s2, s3); }
        }
      }
#elif VC2_UNP_PAIR
      if (--qn == 0) {
        q0 = __builtin_bswap32(n0); q1 = __builtin_bswap32(n1); q2 = __builtin_bswap32(n2); q3 = __builtin_bswap32(n3); qn = 4;
        n0 = m0; n1 = m1; n2 = m2; n3 = m3;
        m0 = o0; m1 = o1; m2 = o2; m3 = o3;
        if ((pairs ^= 1) == 0) { fetch4(pay, m0, m1, m2, m3); fetch4(pay, o0, o1, o2, o3); }
      }
#else
      if (--qn == 0) { q0 = __builtin_bswap32(n0); q1 = __builtin_bswap32(n1); q2 = __builtin_bswap32(n2); q3 = __builtin_bswap32(n3); qn = 4; n0 = m0; n1 = m1; n2 = m2; n3 = m3; fetch4(pay, m0, m1, m2, m3); }
#endif
#else
      if (--qn == 0) { q0 = __builtin_bswap32(n0); q1 = __builtin_bswap32(n1); q2 = __builtin_bswap32(n2); q3 = __builtin_bswap32(n3); qn = 4; fetch4(pay); }
#endif
    }
  }
#endif
};

// Where coefficient j of a component record lives when the finest levels are kept as band planes (BandPlanes):
// level l (0 = finest) holds the last 3 * 4^-l ... of the record; inside it band, block row, column.  Returns the element
// offset from the picture's store of (slice sy, sx)'s coefficient, or -1 for the slice record.
__device__ __forceinline__ int band_plane_at(const BandPlanes &bp, int comp, int n, int j, int sy, int sx, int &lw, int &ow, int *lbase = nullptr) {
  lw = ow = 0;
  if (lbase) *lbase = 0;
  if (j < bp.from[comp]) return -1;
  int l = 0, start = n - (3 << (bp.lbsh[comp][0] + bp.lbsw[comp][0]));
  while (j < start) { ++l; start -= 3 << (bp.lbsh[comp][l] + bp.lbsw[comp][l]); }
  const int lh = bp.lbsh[comp][l], e = j - start;
  lw = bp.lbsw[comp][l]; ow = bp.ow[comp][l];
  if (lbase) *lbase = (int)bp.base[comp][l]; // (byte planes: element e of the plane lives at byte 2 * base + (e - base) = e + base)
  const int b = e >> (lh + lw), rem = e & ((1 << (lh + lw)) - 1), r = rem >> lw, c = rem & ((1 << lw) - 1);
  return (int)bp.base[comp][l] + (b * bp.np[comp][l] + (sy << lh) + r) * ow + (sx << lw) + c; // (a picture's store is < 2^31 elements)
}

#ifndef VC2_UNP16_WAVES
#define VC2_UNP16_WAVES 4
#endif
#ifndef VC2_UNP_BRANCHLESS
#define VC2_UNP_BRANCHLESS 1
#endif
// Eight coefficients of a staging row (four dwords of 16-bit pairs) as the eight bytes of a byte plane (BandPlanes::bytes8).
// A value outside -127 .. 127 leaves the sentinel -128 and its value in the wide array at the element's index `at`; the
// 16-bit sentinel (a value beyond 16 bits) has its wide value already (escape()) and only changes sentinels.
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
// (wide: the picture's wide array; at: the element index of the piece's first coefficient; row2: 0, or -- a piece of two block
// rows of four -- the distance of its second row; the address arithmetic stays in the rare path)
__device__ __forceinline__ uint2 unp_bytes8(int4 v, int32_t *wide, int at, int row2, unsigned long long *stats) {
  auto as2 = [](int x) -> us2_t { union { int i; us2_t u; } c; c.i = x; return c.u; };
  const us2_t k = {127, 127};
  const us2_t m = __builtin_elementwise_max(__builtin_elementwise_max(as2(v.x) + k, as2(v.y) + k), __builtin_elementwise_max(as2(v.z) + k, as2(v.w) + k));
  if (__builtin_expect(max((unsigned)m.x, (unsigned)m.y) > 254u, 0)) {
    if (stats) atomicAdd(stats, 1ull); // (feedback for the next batch's choice of layout: vc2hip_api.hip)
    int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      int32_t *wq = wide + (size_t)at + (row2 && d >= 2 ? row2 - 4 : 0) + 2 * d;
      int lo = vc2_lo16((unsigned)w[d]), hi = vc2_hi16((unsigned)w[d]);
      if ((unsigned)(lo + 127) > 254u) { if (lo != VC2_ST_SENTINEL) wq[0] = lo; lo = -128; }
      if ((unsigned)(hi + 127) > 254u) { if (hi != VC2_ST_SENTINEL) wq[1] = hi; hi = -128; }
      w[d] = (int)vc2_pack16(lo, hi);
    }
    v = make_int4(w[0], w[1], w[2], w[3]);
  }
  // the low bytes of the four 16-bit pairs
  return make_uint2(__builtin_amdgcn_perm((unsigned)v.y, (unsigned)v.x, 0x06040200u), __builtin_amdgcn_perm((unsigned)v.w, (unsigned)v.z, 0x06040200u));
}

template <bool BP8>
__global__ __launch_bounds__(64 * VC2_UNP16_WAVES) void k_hq_unpack16(const UnpackParams p) {
  constexpr int UNP_N = VC2_UNP16_N, UNP_PITCH = UNP_N + 8, PR = UNP_N / 8; // shorts per staging row (8 of slack), 16-byte aligned rows; pieces per row
  __shared__ __attribute__((aligned(16))) short stage[VC2_UNP16_WAVES][64 * UNP_PITCH];
  __shared__ unsigned lut[UNP_LUT_N];
  for (int i = threadIdx.x * 4; i < UNP_LUT_N; i += blockDim.x * 4) *(uint4 *)(lut + i) = *(const uint4 *)(g_unp_lut + i);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Dispatch order (VC2_UNP_ORDER).  2 (the default, round 3's grid): (blocks, pictures, 3) -- the luma blocks of all
  // pictures first, whose wavefronts live twice as long as the chroma ones, then U, then V: the short wavefronts fill the
  // end of the launch.  0: a one-dimensional list that puts the Y, U and V passes over the same 256 slices next to each
  // other on one XCD, so that the payload lines they share are fetched once: rocprofv3
  // FETCH_SIZE 1158 -> 955 MB per launch and 0.587 -> 0.610 ms on the same box -- the traffic is not what bounds this
  // kernel, the order of long and short wavefronts is worth 4 %.  1: component-major inside each picture: 0.593.
  const int pic = blockIdx.y;
  const int nblk = (p.n_slices + 64 * VC2_UNP16_WAVES - 1) / (64 * VC2_UNP16_WAVES);
#ifndef VC2_UNP_ORDER
#define VC2_UNP_ORDER 2
#endif
#if VC2_UNP_ORDER == 0
  const int bx = blockIdx.x & 7, bg = blockIdx.x >> 3;
  const int comp = bg % 3, blk = (bg / 3) * 8 + bx;
#elif VC2_UNP_ORDER == 1 // component-major inside a picture: all its luma blocks, then U, then V
  const int nblk8 = ((nblk + 7) / 8) * 8;
  const int comp = blockIdx.x / nblk8, blk = blockIdx.x % nblk8;
#else // component-major over the launch (grid.z): the luma blocks of all pictures first
  const int comp = blockIdx.z, blk = blockIdx.x;
#endif
  if (blk >= nblk) return; // (the list is padded to whole groups of eight blocks)
  const int slice0 = blk * (64 * VC2_UNP16_WAVES) + wave * 64, slice = slice0 + lane;
  const bool active = slice < p.n_slices;
  const int n = p.comp_n[comp];
  short *st = stage[wave] + lane * UNP_PITCH;
  const uint8_t *pay0 = p.payload + (size_t)pic * p.payload_stride;
  const unsigned mis = (unsigned)((size_t)pay0 & 3);
  const uint8_t *pay = pay0 - mis; // dword aligned: the reader's offsets count from here
  const size_t rec0 = (size_t)pic * p.store_stride + p.comp_off[comp]; // + slice * slice_coefs: a component record
  int32_t *wide = p.store_wide + rec0 + (size_t)(active ? slice : 0) * p.slice_coefs;
  const int sy = (active ? slice : 0) / p.xs, sx = (active ? slice : 0) - sy * p.xs;
  // an escape of coefficient j: the wide element with the index of the store element (record or band plane)
  auto escape = [&](int j, int v) {
    int lw, ow;
    const int at = p.bp.levels ? band_plane_at(p.bp, comp, n, j, sy, sx, lw, ow) : -1;
    if (at >= 0) p.store_wide[(size_t)pic * p.store_stride + (size_t)at] = v;
    else if (j < p.hs.n[comp]) p.store_wide[(size_t)pic * p.store_stride + (size_t)p.hs.base[comp] + (size_t)(active ? slice : 0) * p.hs.n[comp] + j] = v;
    else wide[j] = v;
  };
  Reader32 br;
  {
    unsigned pos = 0, len = 0;
    if (active) {
      const unsigned long long plen = min(p.lens[pic], (unsigned long long)p.payload_stride);
      unsigned long long at = (unsigned long long)p.offsets[(size_t)pic * p.n_slices + slice] + p.prefix;
      auto rd = [&](unsigned long long a) -> unsigned { return a < plen ? pay0[a] : 0u; };
      if (comp == 0) p.qidx[(size_t)pic * p.n_slices + slice] = (int)rd(at);
      at += 1;
      for (int c = 0; c < comp; ++c) at += 1 + (unsigned long long)rd(at) * p.scalar;
      len = rd(at) * p.scalar;
      at += 1;
      if (at + len > plen) {
        atomicOr(p.err, VC2_DEVERR_STREAM);
        len = at < plen ? (unsigned)(plen - at) : 0;
      }
      pos = at < plen ? (unsigned)at : 0u;
    }
#ifdef VC2HIP_ABLATE
    br.dbg_mask = VC2_SKIP(p, 1) ? 0x3FFCu : ~0u;
#endif
    br.safe = (unsigned)min((unsigned long long)p.payload_stride + mis, 0xFFFFFFF0ull);
    br.init(pay, pos + mis, (int)len);
  }
#pragma unroll
  for (int k = 0; k < UNP_PITCH; k += 8) *(int4 *)(st + k) = make_int4(0, 0, 0, 0);
  __syncthreads(); // the table
  int cnt = 0;     // coefficients of the current round already in the row (carried over from the previous one)
  for (int base = 0; base < n; base += UNP_N) {
    const int room = min(UNP_N, n - base);
    for (int turn = 0; cnt < room; ++turn) {
      int used = 0;
      // the look-ups (after two, 13 or more unread bits are left: still a whole index).  Stores go to st[cnt + position]; with no value in the entry they rewrite a zero.
#pragma unroll
      for (int look = 0; look < UNP_LOOKS; ++look) {
        const unsigned w = look ? (unsigned)((br.acc << used) >> 32) : br.top();
        unsigned e = lut[w >> (32 - UNP_LUT_BITS)];
#if VC2_UNP_BRANCHLESS
        // Applied without a branch: an entry of 0 (no whole code in the window) rewrites the zero at st[cnt] and moves
        // nothing, and a lane whose row is full takes 0 for its entry -- three nested EXEC branches per look-up otherwise
        if (look) e = (cnt < room && (look < 3 || used + UNP_LUT_BITS <= br.have)) ? e : 0u; // (a fourth look-up: only with a whole index of unread bits)
#else
        if (e != 0 && (look == 0 || (used != 0 && cnt < room)))
#endif
        {
          st[cnt + (int)((e >> 8) & 15u)] = (short)__builtin_amdgcn_sbfe((int)e, 16, 8);
          st[cnt + (int)((e >> 12) & 15u)] = (short)((int)e >> 24);
          cnt += (int)((e >> 4) & 15u);
          used += (int)(e & 15u);
        }
      }
      // lanes stopped at a code the table does not hold decode that one code
      if ((turn & UNP_LONG_EVERY) == UNP_LONG_EVERY) {
        if (used == 0 && cnt < room) {
          const unsigned hi = br.top();                   // starts with a 0: a code
          const unsigned follow = hi & 0xAAAAAAAAu;       // follow bits sit at even offsets from the code start
          if (follow != 0) {
            const int K = __clz((int)follow) >> 1;        // 1..15 (bit 31 of hi is 0)
            const unsigned body = hi >> ((32 - 2 * K) & 31); // top 2K bits: (0 b) pairs
            const unsigned mag = ((1u << K) | compact_even32(body)) - 1u;
            const int neg = (int)((hi >> ((30 - 2 * K) & 31)) & 1u);
            int v = neg ? (int)(0u - mag) : (int)mag;
            if (!St<int16_t>::fits(v)) { escape(base + cnt, v); v = VC2_ST_SENTINEL; }
            st[cnt++] = (short)v;
            used = 2 * K + 2;
          } else { // longer than 32 bits (outside the reference's domain): bit-serial, wraps like the oracle
            unsigned value = 1;
            for (;;) {
              const unsigned f = br.top() >> 31;
              br.skip(pay, 1);
              if (f) break;
              value = (value << 1) | (br.top() >> 31);
              br.skip(pay, 1);
            }
            value -= 1u;
            int v = 0;
            if (value) {
              v = (br.top() >> 31) ? (int)(0u - value) : (int)value;
              br.skip(pay, 1);
            }
            if (!St<int16_t>::fits(v)) { escape(base + cnt, v); v = VC2_ST_SENTINEL; }
            st[cnt++] = (short)v;
          }
        }
      }
      br.skip(pay, used);
    }
    // flush: UNP_N / 8 lanes x 16 bytes per row (half a 128-byte line of the store).  The staging rows are private to the
    // wavefront, so only its own lanes have to agree (LDS operations of one wavefront execute in order): no workgroup
    // barrier, the four wavefronts of the workgroup drift apart freely.
    wave_lds_sync();
    const short *sw = stage[wave];
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v2i __attribute__((ext_vector_type(2)));
    const int rec_end = VC2_SKIP(p, 2) ? room : min(room, p.bp.from[comp] - base); // coefficients of this round that belong in the slice record
    if (rec_end > 0 && !VC2_SKIP(p, 2)) {
#pragma unroll
      for (int j = 0; j < PR; ++j) {
        const int r = j * (64 / PR) + lane / PR, c = (lane % PR) * 8;
        if (slice0 + r < p.n_slices && c < rec_end) {
          const int4 v = *(const int4 *)(sw + r * UNP_PITCH + c);
          const v4i vv = {v.x, v.y, v.z, v.w};
          int16_t *dst = (int16_t *)p.store + rec0 + (size_t)(slice0 + r) * p.slice_coefs + base + c;
          if (base + c < p.hs.n[comp]) // a piece of the record's head (HeadSplit: the deep levels' coefficients of all slices side by side)
            dst = (int16_t *)p.store + (size_t)pic * p.store_stride + (size_t)p.hs.base[comp] + (size_t)(slice0 + r) * p.hs.n[comp] + base + c;
          __builtin_nontemporal_store(vv, (__attribute__((address_space(1))) v4i *)(size_t)dst);
        }
      }
    }
    // band planes: neighbouring lanes are neighbouring slices, and their pieces of a band row are neighbours in the
    // plane.  A block row of 8 coefficients is one 16-byte piece per lane; of 4, two 8-byte halves (two rows); of 16 or
    // more, 2^(lw-3) pieces per slice: the lanes then take the pieces of the row in plane order (piece t of the run of
    // 64 slices = slice t / P, piece t % P -- read from that slice's staging row), so that every store instruction
    // still writes one contiguous kilobyte.  (The piece index is wave-uniform: level / band / row on the SALU.)
    if (rec_end < room) {
      for (int q = 0; q < PR;) {
        const int j0 = base + 8 * q;
        if (8 * q >= room) break;
        if (j0 < p.bp.from[comp]) { ++q; continue; }
        int lw, ow, lbase;
        int at = band_plane_at(p.bp, comp, n, j0, sy, sx, lw, ow, &lbase); // (lw, ow, lbase: the same in every lane)
        if (!active) at = -1;
        lw = __builtin_amdgcn_readfirstlane(lw);
        const int lp = lw - 3, P = lw >= 4 ? 1 << lp : 1; // pieces per block row
        if (lw >= 4 && q + P <= PR && ((j0 >> 3) & (P - 1)) == 0) { // (a whole row inside the round; else piece by piece)
          for (int k = 0; k < P; ++k) {
            const int t = lane + 64 * k, sl = t >> lp, pp = t & (P - 1);
            const int at_t = __shfl(at, sl);
            if (at_t >= 0) {
              const int4 v = *(const int4 *)(sw + sl * UNP_PITCH + 8 * (q + pp));
              if constexpr (BP8) {
                const uint2 b8 = unp_bytes8(v, p.store_wide + (size_t)pic * p.store_stride, at_t + 8 * pp, 0, p.stats);
                const v2i vb = {(int)b8.x, (int)b8.y};
                char *dst = (char *)((int16_t *)p.store + (size_t)pic * p.store_stride) + (size_t)(at_t + 8 * pp) + (size_t)lbase;
                __builtin_nontemporal_store(vb, (__attribute__((address_space(1))) v2i *)(size_t)dst);
              } else {
              const v4i vv = {v.x, v.y, v.z, v.w};
              int16_t *dst = (int16_t *)p.store + (size_t)pic * p.store_stride + (size_t)(at_t + 8 * pp);
              __builtin_nontemporal_store(vv, (__attribute__((address_space(1))) v4i *)(size_t)dst);
              }
            }
          }
          q += P;
          continue;
        }
        if (BP8 && at >= 0) {
          const int4 v = *(const int4 *)(st + 8 * q);
          char *dst = (char *)((int16_t *)p.store + (size_t)pic * p.store_stride) + (size_t)at + (size_t)lbase;
          const uint2 b8 = unp_bytes8(v, p.store_wide + (size_t)pic * p.store_stride, at, lw >= 3 ? 0 : ow, p.stats);
          if (lw >= 3) {
            const v2i vb = {(int)b8.x, (int)b8.y};
            __builtin_nontemporal_store(vb, (__attribute__((address_space(1))) v2i *)(size_t)dst);
          } else { // two rows of 4
            __builtin_nontemporal_store((int)b8.x, (__attribute__((address_space(1))) int *)(size_t)dst);
            __builtin_nontemporal_store((int)b8.y, (__attribute__((address_space(1))) int *)(size_t)(dst + ow));
          }
        } else if (at >= 0) {
          int16_t *dst = (int16_t *)p.store + (size_t)pic * p.store_stride + (size_t)at;
          const int4 v = *(const int4 *)(st + 8 * q);
          if (lw >= 3) {
            const v4i vv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(vv, (__attribute__((address_space(1))) v4i *)(size_t)dst);
          } else { // two rows of 4 (the host admits no narrower blocks)
            const v2i lo = {v.x, v.y}, hi = {v.z, v.w};
            __builtin_nontemporal_store(lo, (__attribute__((address_space(1))) v2i *)(size_t)dst);
            __builtin_nontemporal_store(hi, (__attribute__((address_space(1))) v2i *)(size_t)(dst + ow));
          }
        }
        ++q;
      }
    }
    wave_lds_sync();
    // coefficients decoded beyond the round move to the front of the row
    const int4 spill = *(const int4 *)(st + UNP_N);
#pragma unroll
    for (int k = 0; k < UNP_PITCH; k += 8) *(int4 *)(st + k) = make_int4(0, 0, 0, 0);
    cnt = max(cnt - UNP_N, 0);
    if (cnt) *(int4 *)st = spill;
  }
}

void vc2_launch_unpack(Launcher &L, const UnpackParams &p0, int n_pictures, hipStream_t s) {
  UnpackParams p = p0;
#ifdef VC2HIP_ABLATE
  { const char *e = getenv("VC2HIP_DEBUG_UNPACK"); p.debug_skip = e ? atoi(e) : 0; }
#endif
  if (!p.bp.levels) for (int c = 0; c < 3; ++c) p.bp.from[c] = 1 << 30; // everything in the slice records
  if (p.xs < 1) p.xs = 1;
  vc2_prof_begin(L, "hq_unpack", s);
  if (p.store16) {
    constexpr int T = 64 * VC2_UNP16_WAVES;
    const int nblk = (p.n_slices + T - 1) / T;
    const bool b8 = p.bp.levels && p.bp.bytes8;
#if VC2_UNP_ORDER == 2
    if (b8) VC2_LAUNCH(L, k_hq_unpack16<true>, dim3(nblk, n_pictures, 3), dim3(T), 0, s, p);
    else VC2_LAUNCH(L, k_hq_unpack16<false>, dim3(nblk, n_pictures, 3), dim3(T), 0, s, p);
#else
    if (b8) VC2_LAUNCH(L, k_hq_unpack16<true>, dim3(((nblk + 7) / 8) * 8 * 3, n_pictures), dim3(T), 0, s, p);
    else VC2_LAUNCH(L, k_hq_unpack16<false>, dim3(((nblk + 7) / 8) * 8 * 3, n_pictures), dim3(T), 0, s, p);
#endif
    vc2_prof_end(L, s);
    return;
  }
  // A/B on MI355X, 16 UHD pictures: whole 128-byte lines (32 coefficients per round) 0.48 ms, 64-byte runs 0.52 ms (the other
  // half of every line is fetched back: FETCH_SIZE ~1 GB for 0.15 GB of payload), plain instead of non-temporal stores 0.69 ms
  static const int wide = vc2_tune_int("VC2HIP_UNPACK_WIDE", 1);
  static const int nt = vc2_tune_int("VC2HIP_UNPACK_NT", 1);
  if (!nt) VC2_LAUNCH(L, (k_hq_unpack<16, false>), dim3((p.n_slices + 255) / 256, n_pictures, 3), dim3(256), 0, s, p);
  else if (wide) VC2_LAUNCH(L, k_hq_unpack<32>, dim3((p.n_slices + 255) / 256, n_pictures, 3), dim3(256), 0, s, p);
  else VC2_LAUNCH(L, k_hq_unpack<16>, dim3((p.n_slices + 255) / 256, n_pictures, 3), dim3(256), 0, s, p);
  vc2_prof_end(L, s);
}

// ------------------------------------------------------------------------------------------
// slice index: start offset of every slice of a VBR picture payload.
// The offsets form a serial chain through the length bytes; it is cut into CH-byte chunks:
//   1. tables : per chunk, for every possible entry offset e, walk the chain to the chunk end
//               -> (exit offset into the next chunk, slices started)       [parallel in e, chunk]
//   2. chain  : per picture, follow entry -> exit through the chunk tables  [one lane/picture]
//   3. emit   : per chunk, walk again from the now-known entry and write the offsets
// ------------------------------------------------------------------------------------------
// chunk size: 16 KiB, or 32 KiB when a slice can be longer than 8191 bytes (entry offsets must stay below the
// chunk size and chunk-relative positions below 2^16)
// (8 KiB while an entry region and its landing region fit one: four workgroups per CU instead of two, half the hops per
// walk -- the table kernel's time is its chain of dependent LDS reads)
// (measured, same box, table + chain + emit: scalar 1 (E = 769, 1080p) 0.229 -> 0.203 ms with 8 KiB chunks; scalar 2 (E = 1534, UHD)
// 0.248 -> 0.254 ms: the E entries and the table of a chunk do not shrink with it)
static int idx_chunk(int E) { return E <= 1024 ? 8192 : E <= 8191 ? 16384 : 32768; }

__device__ __forceinline__ int slice_len_lds(const uint8_t *b, int pos, int prefix, int scalar) {
  int q = pos + prefix + 1;
  q += 1 + b[q] * scalar;
  q += 1 + b[q] * scalar;
  q += 1 + b[q] * scalar;
  return q - pos;
}

__device__ void stage_chunk(uint8_t *lds, const uint8_t *pay, unsigned long long plen,
                            unsigned long long c0, int nbytes) {
  // nbytes is a multiple of 16; payload slots and chunk starts are 16-byte aligned
  for (int i = threadIdx.x * 16; i < nbytes; i += blockDim.x * 16) {
    uint4 v = make_uint4(0, 0, 0, 0);
    const unsigned long long a = c0 + i;
    if (a + 16 <= plen) v = *(const uint4 *)(pay + a);
    else {
      unsigned w[4] = {0, 0, 0, 0};
      for (int k = 0; k < 16; ++k) if (a + k < plen) w[k >> 2] |= (unsigned)pay[a + k] << (8 * (k & 3));
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *(uint4 *)(lds + i) = v;
  }
}

static constexpr int IDX_MAX_E = 32767;  // beyond: serial walk (k_index_serial)
constexpr int idx_threads(int CH, int GS = 0) { return (CH >> GS) <= 8192 ? 512 : 1024; } // by the chunk's candidate positions (GS: k_index_tables_nx)

// Chunk function by table walk.  One pass fills next[i] = the position reached if a slice started at byte i
// of the chunk, from the three length bytes behind every byte position (throughput-bound LDS work: PER
// independent reads per thread and step).  Then every entry offset walks next[] -- one LDS read per hop
// instead of three dependent ones -- until it leaves the chunk.
// A table entry is one word: exit offset << 16 | slices (both below 2^15: E <= IDX_MAX_E, a chunk holds at most CH / 4 slices).
// (Per-phase stamps of a workgroup's life, 8 KiB chunks, scalar 2: chunk in 1.9 us, next[] 1.8, entry walks 1.2, landing
// walks 0.6, table out ~2: a chain of latencies at ~55 % of the CU's LDS instruction rate.  Tried on it and measured no
// better: next[] in place of the staged bytes with 256-thread workgroups (six chunks per CU in flight, but next[] alone then
// took 5 us: LDS-bound), a thread's walks advanced side by side, a persistent grid that takes the chunks from the
// lengths and fetches the next chunk's bytes while it works (same time at 8 KiB, 0.24 against 0.16 ms at 16 KiB),
// the workgroups of chunks beyond the payload last in the grid instead of between those of every picture.)
// GS: every slice of the picture starts at a multiple of G = 2^GS bytes -- a slice is prefix + 4 + scalar * (three length
// bytes) long, so with G = the largest power of two (up to 4) in gcd(scalar, prefix + 4) every slice length and, from
// offset 0, every slice start is a multiple of G whatever the bytes say.  Only those positions can start a slice: next[],
// the entry offsets, the landing bytes and the tables are kept per G bytes -- half (scalar 2: cfg 2 / 3) or a quarter
// (scalar 8: cfg 4) of the LDS reads, walks and table words of round 3 (0.31 -> 0.2 ms per 32 UHD pictures).
template <int CH, int GS>
__global__ __launch_bounds__(idx_threads(CH, GS)) void k_index_tables_nx(const uint8_t *payload, long long stride,
                                                                 const unsigned long long *lens, unsigned *tables,
                                                                 int n_chunks, int E, int prefix, int scalar, unsigned *err, const unsigned *skip,
                                                                 int merge) {
  constexpr int NT = idx_threads(CH, GS), NP = CH >> GS; // candidate positions of a chunk
  if (skip && *skip == 0) return; // the HQ_CBR short cut found every slice where the byte budgets put it (k_cbr_index_check)
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_b[];
  const int chunk = blockIdx.x, pic = blockIdx.y;
  // a length beyond the picture's slot (hostile or uninitialised) is an error and is never followed out of the slot
  if (chunk == 0 && threadIdx.x == 0 && lens[pic] > (unsigned long long)stride) atomicOr(err, VC2_DEVERR_STREAM);
  const unsigned long long plen = min(lens[pic], (unsigned long long)stride), c0 = (unsigned long long)chunk * CH;
  if (c0 >= plen) return;
  const int nbytes = (CH + E + 16 + 15) & ~15;
  const int EU = E >> GS;                                  // entry offsets (E is a multiple of G)
  unsigned short *nx = (unsigned short *)(lds_b + nbytes); // per candidate position: the byte reached; >= CH: left the chunk at offset nx - CH
  stage_chunk(lds_b, payload + (size_t)pic * stride, plen, c0, nbytes);
  __syncthreads();
  const int lim = (int)min((unsigned long long)CH, plen - c0); // never walk the zero fill behind the payload
  constexpr int PER = NP / NT;
  {
    int q[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) q[k] = ((threadIdx.x + k * NT) << GS) + prefix + 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int len[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) len[k] = lds_b[min(q[k], nbytes - 1)];
#pragma unroll
      for (int k = 0; k < PER; ++k) q[k] += 1 + len[k] * scalar;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int n = q[k];
      // a slice that ends behind the payload end leaves the chunk at offset 0 unless it really reaches the next chunk
      nx[threadIdx.x + k * NT] = (unsigned short)(n >= lim ? (n >= CH ? n : CH) : n);
    }
  }
  __syncthreads();
  unsigned *tab = tables + ((size_t)pic * n_chunks + chunk) * EU;
  if (!merge) {
    for (int e = threadIdx.x; e < EU; e += NT) {
      int pos = e << GS, cnt = 0;
      if (pos >= lim) pos = CH; // starts behind the payload end: no slice
      while (pos < CH) { pos = nx[pos >> GS]; ++cnt; }
      tab[e] = (unsigned)(pos - CH) << 16 | (unsigned)cnt;
    }
    return;
  }
  // The E walks merge: two of them that ever stand on the same byte stay together, and after their first hops out of
  // the entry region [0, E) most stand on one of a few bytes of [E, 2E).  So every entry walks only until it has left
  // [0, E); the distinct landing bytes are collected, walked once each to the chunk's end, and every entry adds its own
  // hops to its landing byte's result -- a fraction of the dependent LDS reads of E full walks (the kernel's time is
  // those reads).
  unsigned *W = (unsigned *)(nx + NP);               // per candidate position of [E, 2E): the result of a walk from it (exit offset << 16 | slices)
  unsigned *need = W + EU;                           // one bit per candidate position of [E, 2E): some walk landed here
  unsigned short *list = (unsigned short *)(need + (EU + 31) / 32);  // the landing positions, dense
  __shared__ unsigned s_count;
  for (int t = threadIdx.x; t < (EU + 31) / 32; t += NT) need[t] = 0;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  constexpr int EPT = 4096 / NT; // entries per thread (the launcher merges only entry regions of up to 4096 offsets)
  int la[EPT], ca[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = threadIdx.x + k * NT;
    la[k] = CH; ca[k] = 0;
    if (e < EU && (e << GS) < lim) {
      int pos = e << GS, cnt = 0;
      while (pos < E) { pos = nx[pos >> GS]; ++cnt; }
      la[k] = pos; ca[k] = cnt;
      if (pos < CH) {
        const int t = (pos - E) >> GS;
        if (!(atomicOr(&need[t >> 5], 1u << (t & 31)) >> (t & 31) & 1u)) list[atomicAdd(&s_count, 1u)] = (unsigned short)t;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (int)s_count; i += NT) {
    const int t = list[i];
    int pos = E + (t << GS), cnt = 0;
    while (pos < CH) { pos = nx[pos >> GS]; ++cnt; }
    W[t] = (unsigned)(pos - CH) << 16 | (unsigned)cnt;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = threadIdx.x + k * NT;
    if (e >= EU) continue;
    if (la[k] >= CH) tab[e] = (unsigned)(la[k] - CH) << 16 | (unsigned)ca[k];
    else tab[e] = W[(la[k] - E) >> GS] + (unsigned)ca[k];
  }
}

// The chain through the chunk functions is followed in three levels so that only n_chunks / IDX_GROUP^2 + 2 IDX_GROUP
// dependent table reads are serial instead of n_chunks (each one a trip to L2 or beyond, ~0.8 us: a UHD-2 picture of 2400
// chunks took 0.16 ms with two levels):
//   group  : compose the functions of IDX_GROUP consecutive chunks, for every entry offset (parallel); then the functions
//            of IDX_GROUP consecutive groups the same way (super-groups)
//   chain  : per picture, follow entry -> exit through the super-groups, expand every super-group into its groups and
//            every group into its chunks
static constexpr int IDX_GROUP = 16;

// an entry of a function table: (exit offset, slices).  WORD: the chunk tables' packed form exit << 16 | slices
template <bool WORD> __device__ __forceinline__ uint2 idx_entry(const void *tab, size_t i) {
  if constexpr (WORD) { const unsigned t = ((const unsigned *)tab)[i]; return make_uint2(t >> 16, t & 0xFFFFu); }
  else return ((const uint2 *)tab)[i];
}

// out[g] = in[16 g + 15] o ... o in[16 g]; `span` = payload bytes one input function covers
// (E: entry offsets per table = bytes >> gsh, see k_index_tables_nx; the exit offsets inside the entries stay bytes)
template <bool WORD>
__global__ __launch_bounds__(256) void k_index_group(const unsigned long long *lens, long long stride, const void *tables,
                                                     uint2 *groups, int n_chunks, int n_groups, int E, long long span, const unsigned *skip,
                                                     int dedupe, int gsh) {
  if (skip && *skip == 0) return; // the HQ_CBR short cut found every slice where the byte budgets put it (k_cbr_index_check)
  extern __shared__ __attribute__((aligned(8))) unsigned lds_g[];
  const int g = blockIdx.x, pic = blockIdx.y;
  const unsigned long long plen = min(lens[pic], (unsigned long long)stride);
  if ((unsigned long long)g * IDX_GROUP * (unsigned long long)span >= plen) return;
  if (!dedupe) {
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
      unsigned x = (unsigned)e << gsh, cnt = 0;
      for (int k = 0; k < IDX_GROUP; ++k) {
        const int c = g * IDX_GROUP + k;
        if (c >= n_chunks || (unsigned long long)c * (unsigned long long)span >= plen) break;
        const uint2 t = idx_entry<WORD>(tables, ((size_t)pic * n_chunks + c) * E + (x >> gsh));
        x = t.x;
        cnt += t.y;
      }
      groups[((size_t)pic * n_groups + g) * E + e] = make_uint2(x, cnt);
    }
    return;
  }
  // The E entries leave the group's first chunk at only a few distinct offsets (chains that meet stay together): the
  // other fifteen chunks are followed once per distinct offset instead of once per entry.
  unsigned *need = lds_g;                                // per offset: some entry leaves the first chunk here
  uint2 *res = (uint2 *)(need + ((E + 1) & ~1));         // its way through the rest of the group
  unsigned short *list = (unsigned short *)(res + E);    // the distinct offsets, dense
  __shared__ unsigned s_count;
  for (int e = threadIdx.x; e < E; e += blockDim.x) need[e] = 0;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  const int c0 = g * IDX_GROUP;
  const size_t t0 = ((size_t)pic * n_chunks + c0) * E;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const unsigned x = idx_entry<WORD>(tables, t0 + e).x >> gsh;
    if (atomicExch(&need[x], 1u) == 0u) list[atomicAdd(&s_count, 1u)] = (unsigned short)x;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (int)s_count; i += blockDim.x) {
    unsigned x = list[i], cnt = 0; // (entry units here; the tables' exits are bytes)
    const unsigned x0 = x;
    for (int k = 1; k < IDX_GROUP; ++k) {
      const int c = c0 + k;
      if (c >= n_chunks || (unsigned long long)c * (unsigned long long)span >= plen) break;
      const uint2 t = idx_entry<WORD>(tables, ((size_t)pic * n_chunks + c) * E + x);
      x = t.x >> gsh;
      cnt += t.y;
    }
    res[x0] = make_uint2(x << gsh, cnt);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const uint2 t = idx_entry<WORD>(tables, t0 + e);
    const uint2 r = res[t.x >> gsh];
    groups[((size_t)pic * n_groups + g) * E + e] = make_uint2(r.x, t.y + r.y);
  }
}

constexpr int IDX_MAX_GROUPS = 1024;
__global__ __launch_bounds__(1024) void k_index_chain(const unsigned long long *lens, long long stride, const unsigned *tables,
                                                    const uint2 *groups, const uint2 *supers, uint2 *entries, int n_chunks,
                                                    int n_groups, int n_supers, int E, int IDX_CH, const unsigned *skip, int gsh) {
  if (skip && *skip == 0) return; // the HQ_CBR short cut found every slice where the byte budgets put it (k_cbr_index_check)
  __shared__ uint2 g_entry[IDX_MAX_GROUPS];              // the launcher keeps n_groups within it (larger chunks for larger slots)
  __shared__ uint2 s_entry[IDX_MAX_GROUPS / IDX_GROUP];
  const int pic = blockIdx.x;
  const unsigned long long plen = min(lens[pic], (unsigned long long)stride);
  const unsigned long long gspan = (unsigned long long)IDX_GROUP * IDX_CH, sspan = gspan * IDX_GROUP;
  if (n_supers == 0) { // few groups (the launcher's choice): they are followed directly
    if (threadIdx.x == 0) {
      unsigned entry = 0, base = 0;
      for (int g = 0; g < n_groups; ++g) {
        g_entry[g] = make_uint2(entry, base);
        if ((unsigned long long)g * gspan >= plen) continue;
        const uint2 t = groups[((size_t)pic * n_groups + g) * E + (entry >> gsh)];
        entry = t.x;
        base += t.y;
      }
    }
  } else if (threadIdx.x == 0) { // the only serial part: one read per super-group
    unsigned entry = 0, base = 0;
    for (int sg = 0; sg < n_supers; ++sg) {
      s_entry[sg] = make_uint2(entry, base);
      if ((unsigned long long)sg * sspan >= plen) continue;
      const uint2 t = supers[((size_t)pic * n_supers + sg) * E + (entry >> gsh)];
      entry = t.x;
      base += t.y;
    }
  }
  __syncthreads();
  for (int sg = threadIdx.x; sg < n_supers; sg += blockDim.x) { // every super-group into its groups
    unsigned entry = s_entry[sg].x, base = s_entry[sg].y;
    for (int k = 0; k < IDX_GROUP; ++k) {
      const int g = sg * IDX_GROUP + k;
      if (g >= n_groups) break;
      g_entry[g] = make_uint2(entry, base);
      if ((unsigned long long)g * gspan >= plen) continue;
      const uint2 t = groups[((size_t)pic * n_groups + g) * E + (entry >> gsh)];
      entry = t.x;
      base += t.y;
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < n_groups; g += blockDim.x) { // every group into its chunks
    unsigned entry = g_entry[g].x, base = g_entry[g].y;
    for (int k = 0; k < IDX_GROUP; ++k) {
      const int c = g * IDX_GROUP + k;
      if (c >= n_chunks) break;
      entries[(size_t)pic * n_chunks + c] = make_uint2(entry, base);
      if ((unsigned long long)c * IDX_CH >= plen) continue;
      const unsigned t = tables[((size_t)pic * n_chunks + c) * E + (entry >> gsh)];
      entry = t >> 16;
      base += t & 0xFFFFu;
    }
  }
}

// One lane per chunk walks from the chunk's (now known) entry and writes the offsets: three dependent byte loads per
// slice, served by L2 -- a chunk's lane needs ~60 of its 8192 bytes, and all chunks of the batch walk at the same time
// (round 2 staged every chunk whole into LDS for one walking lane: eight chunks per CU at a time, 0.047 ms).
__global__ __launch_bounds__(64) void k_index_emit(const uint8_t *payload, long long stride,
                                                   const unsigned long long *lens, const uint2 *entries,
                                                   uint32_t *offsets, int n_chunks, int E, int n_slices,
                                                   int prefix, int scalar, unsigned *err, int IDX_CH, const unsigned *skip) {
  if (skip && *skip == 0) return; // the HQ_CBR short cut found every slice where the byte budgets put it (k_cbr_index_check)
  (void)E;
  const int chunk = blockIdx.x * 64 + threadIdx.x, pic = blockIdx.y;
  if (chunk >= n_chunks) return;
  const unsigned long long plen = min(lens[pic], (unsigned long long)stride), c0 = (unsigned long long)chunk * IDX_CH;
  if (c0 >= plen) return;
  const uint2 en = entries[(size_t)pic * n_chunks + chunk];
  if ((int)en.y >= n_slices || (int)en.x >= IDX_CH) return;
  const uint8_t *pay = payload + (size_t)pic * stride;
  unsigned long long pos = c0 + en.x;
  const unsigned long long end = c0 + IDX_CH;
  int k = (int)en.y;
  while (pos < end && k < n_slices) {
    offsets[(size_t)pic * n_slices + k] = (uint32_t)pos;
    if (pos >= plen) { atomicOr(err, VC2_DEVERR_STREAM); break; } // (what follows starts behind the payload too)
    unsigned long long q = pos + prefix + 1;
    for (int c = 0; c < 3; ++c) q += 1 + (q < plen ? (unsigned long long)pay[q] * scalar : 0ull);
    pos = q;
    ++k;
  }
}

static int idx_entries(int prefix, int scalar) { return prefix + 4 + 3 * 255 * scalar; }
// log2 of G: every slice starts at a multiple of G bytes (k_index_tables_nx)
static int idx_gsh(int prefix, int scalar) {
  int gsh = 0;
  while (gsh < 2 && scalar % (2 << gsh) == 0 && (prefix + 4) % (2 << gsh) == 0) ++gsh;
  return gsh;
}
#ifndef VC2_IDX_CHUNK_BY_UNITS
#define VC2_IDX_CHUNK_BY_UNITS 0
#endif
static int idx_chunk_of(int prefix, int scalar) { const int E = idx_entries(prefix, scalar); return idx_chunk(VC2_IDX_CHUNK_BY_UNITS ? std::max(E >> idx_gsh(prefix, scalar), (E + 3) / 4) : E); }

// Last resort for slices that can exceed 32767 bytes (slice size scalar > 42): one lane per picture follows the
// length bytes through memory.  Correct for any stream, three dependent loads per slice.
__global__ __launch_bounds__(64) void k_index_serial(const uint8_t *payload, long long stride, const unsigned long long *lens,
                                                     uint32_t *offsets, int n_slices, int prefix, int scalar, unsigned *err,
                                                     int n_pictures, const unsigned *skip) {
  if (skip && *skip == 0) return; // the HQ_CBR short cut found every slice where the byte budgets put it (k_cbr_index_check)
  const int pic = blockIdx.x * 64 + threadIdx.x;
  if (pic >= n_pictures) return;
  const uint8_t *pay = payload + (size_t)pic * stride;
  if (lens[pic] > (unsigned long long)stride) atomicOr(err, VC2_DEVERR_STREAM);
  const unsigned long long plen = min(lens[pic], (unsigned long long)stride);
  unsigned long long pos = 0;
  for (int k = 0; k < n_slices; ++k) {
    offsets[(size_t)pic * n_slices + k] = (uint32_t)pos;
    if (pos >= plen) { atomicOr(err, VC2_DEVERR_STREAM); continue; }
    unsigned long long q = pos + prefix + 1;
    for (int c = 0; c < 3; ++c) q += 1 + (q < plen ? (unsigned long long)pay[q] * scalar : 0ull);
    pos = q;
  }
}

// HQ_CBR pictures: every slice fills its byte budget exactly (Slices.cpp:352-368: the last component absorbs the
// remainder), so the slice offsets of a conforming stream are the running sums of the budgets -- no chain to follow.
// The decoder checks that claim against the stream itself: at every predicted offset the three length bytes must add up
// to that slice's budget and the picture must be as long as all budgets together.  By induction over the slices that is
// exactly "the chain through the length bytes visits these offsets"; one mismatch anywhere (*bad != 0) and the general
// index kernels run as for any other stream (they return at once otherwise).
__global__ __launch_bounds__(256) void k_cbr_index_check(const uint8_t *payload, long long stride, const unsigned long long *lens,
                                                         const int32_t *budget, const uint32_t *cbr_offs, unsigned long long total,
                                                         int n_slices, int prefix, int scalar, unsigned *bad) {
  const int slice = blockIdx.x * 256 + threadIdx.x, pic = blockIdx.y;
  if (slice >= n_slices) return;
  const uint8_t *pay = payload + (size_t)pic * stride;
  const unsigned long long plen = min(lens[pic], (unsigned long long)stride);
  bool ok = lens[pic] == total;
  unsigned long long pos = (unsigned long long)cbr_offs[slice] + prefix + 1;
  const unsigned long long end = (unsigned long long)cbr_offs[slice] + prefix + (unsigned)budget[slice];
  for (int c = 0; c < 3 && ok; ++c) {
    ok = pos < plen;
    if (ok) pos += 1 + (unsigned long long)pay[pos] * scalar;
  }
  if (!ok || pos != end || end > plen) atomicOr(bad, 1u);
}
__global__ __launch_bounds__(256) void k_cbr_index_fill(const uint32_t *cbr_offs, uint32_t *offsets, int n_slices, const unsigned *bad) {
  const int slice = blockIdx.x * 256 + threadIdx.x, pic = blockIdx.y;
  if (slice < n_slices && *bad == 0) offsets[(size_t)pic * n_slices + slice] = cbr_offs[slice];
}
void vc2_launch_cbr_index(Launcher &L, const uint8_t *payload, long long stride, const unsigned long long *lens, const int32_t *budget,
                          const uint32_t *cbr_offs, unsigned long long total, uint32_t *offsets, int n_slices, int prefix, int scalar,
                          int n_pictures, unsigned *bad, hipStream_t s) {
  vc2_prof_begin(L, "slice_index_cbr", s);
  const dim3 grid((n_slices + 255) / 256, n_pictures);
  VC2_LAUNCH(L, k_cbr_index_check, grid, dim3(256), 0, s, payload, stride, lens, budget, cbr_offs, total, n_slices, prefix, scalar, bad);
  VC2_LAUNCH(L, k_cbr_index_fill, grid, dim3(256), 0, s, cbr_offs, offsets, n_slices, bad);
  vc2_prof_end(L, s);
}

bool vc2_slice_index_supported(int prefix, int scalar);
bool vc2_slice_index_supported(int prefix, int scalar) { (void)prefix; (void)scalar; return true; }

size_t vc2_slice_index_workspace(int n_pictures, size_t max_payload, int prefix, int scalar) {
  const size_t E = idx_entries(prefix, scalar);
  if (E > (size_t)IDX_MAX_E) return 256;
  const size_t ch = (size_t)idx_chunk_of(prefix, scalar);
  const size_t n_chunks = (max_payload + ch - 1) / ch + 1;
  if (n_chunks > (size_t)1024 * 16) return 256; // serial walk (see vc2_launch_slice_index)
  const size_t n_groups = (n_chunks + IDX_GROUP - 1) / IDX_GROUP;
  const size_t n_supers = (n_groups + IDX_GROUP - 1) / IDX_GROUP;
  return (size_t)n_pictures * (n_chunks * E * sizeof(unsigned) + (n_chunks + (n_groups + n_supers) * E) * sizeof(uint2)) + 512;
}

void vc2_launch_slice_index(Launcher &L, const uint8_t *payload, long long payload_stride,
                            const unsigned long long *lens, uint32_t *offsets, int n_slices,
                            int prefix, int scalar, int n_pictures, unsigned *err, hipStream_t s,
                            void *workspace, size_t workspace_bytes, const unsigned *skip) {
  const int E = idx_entries(prefix, scalar);
  (void)workspace_bytes;
  // the chain kernel holds one entry per group of 16 chunks in LDS: slots beyond IDX_MAX_GROUPS groups (256 MiB at
  // 16 KiB chunks) take the serial walk like over-long slices do
  const bool too_many = E <= IDX_MAX_E &&
      ((size_t)payload_stride + idx_chunk_of(prefix, scalar) - 1) / idx_chunk_of(prefix, scalar) + 1 > (size_t)IDX_MAX_GROUPS * IDX_GROUP;
  if (E > IDX_MAX_E || too_many) {
    vc2_prof_begin(L, "slice_index_serial", s);
    VC2_LAUNCH(L, k_index_serial, dim3((n_pictures + 63) / 64), dim3(64), 0, s, payload, payload_stride, lens,
                       offsets, n_slices, prefix, scalar, err, n_pictures, skip);
    vc2_prof_end(L, s);
    return;
  }
  const int ch = idx_chunk_of(prefix, scalar);
  // every slice starts at a multiple of G (k_index_tables_nx): the tables hold one entry per G bytes
  const int gsh = idx_gsh(prefix, scalar);
  const int EU = E >> gsh; // (E = prefix + 4 + 765 * scalar is a multiple of G)
  // chunk count is bounded by the payload slot size (lens live on the device)
  const int n_chunks = (int)(((size_t)payload_stride + ch - 1) / ch) + 1;
  unsigned *tables = (unsigned *)workspace;
  uint2 *entries = (uint2 *)(((size_t)(tables + (size_t)n_pictures * n_chunks * EU) + 15) & ~(size_t)15);
  const size_t stage_bytes = (size_t)((ch + E + 16 + 15) & ~15);
  vc2_prof_begin(L, "slice_index_tables", s);
  {
    // merged walks (see the kernel): entry regions of up to 4096 offsets whose landing region lies inside the chunk
    static const bool no_merge = vc2_tune_int("VC2HIP_IDX_NO_MERGE", 0) != 0;
    const int merge = !no_merge && EU <= 4096 && 2 * E <= ch;
    const size_t lds = stage_bytes + (size_t)(ch >> gsh) * 2 + (merge ? (size_t)EU * 6 + ((size_t)(EU + 31) / 32) * 4 + 16 : 0);
#define VC2_IDX_TABLES(CH, GS)                                                                                                    \
  do {                                                                                                                        \
    vc2_allow_lds((const void *)k_index_tables_nx<CH, GS>, lds);                                                                  \
    VC2_LAUNCH(L, (k_index_tables_nx<CH, GS>), dim3(n_chunks, n_pictures), dim3(idx_threads(CH, GS)), lds, s, payload, payload_stride, \
               lens, tables, n_chunks, E, prefix, scalar, err, skip, merge);                                                  \
  } while (0)
#define VC2_IDX_TABLES_G(CH) do { if (gsh == 0) VC2_IDX_TABLES(CH, 0); else if (gsh == 1) VC2_IDX_TABLES(CH, 1); else VC2_IDX_TABLES(CH, 2); } while (0)
    if (ch == 8192) VC2_IDX_TABLES_G(8192);
    else if (ch == 16384) VC2_IDX_TABLES_G(16384);
    else VC2_IDX_TABLES_G(32768);
#undef VC2_IDX_TABLES_G
#undef VC2_IDX_TABLES
  }
  vc2_prof_end(L, s);
  const int n_groups = (n_chunks + IDX_GROUP - 1) / IDX_GROUP; // <= 1024 (g_entry)
  // (the third level costs a launch: 5 us; it pays from a few hundred groups on -- UHD-2 pictures: 0.163 -> 0.132 ms)
  const int n_supers = n_groups > 128 ? (n_groups + IDX_GROUP - 1) / IDX_GROUP : 0;
  uint2 *groups = entries + (size_t)n_pictures * n_chunks;
  uint2 *supers = groups + (size_t)n_pictures * n_groups * EU;
  vc2_prof_begin(L, "slice_index_chain", s);
  {
    // (exit offsets are below E and fit 16 bits up to IDX_MAX_E; the dedupe tables need 14 bytes of LDS per entry)
    static const bool no_dedupe = vc2_tune_int("VC2HIP_IDX_NO_DEDUPE", 0) != 0;
    const int dedupe = !no_dedupe && EU <= 8192;
    const size_t glds = dedupe ? ((size_t)((EU + 1) & ~1) * 4 + (size_t)EU * 8 + (size_t)EU * 2 + 16) : 0;
    if (glds) { vc2_allow_lds((const void *)k_index_group<true>, glds); vc2_allow_lds((const void *)k_index_group<false>, glds); }
    VC2_LAUNCH(L, k_index_group<true>, dim3(n_groups, n_pictures), dim3(256), glds, s, lens, payload_stride, (const void *)tables, groups, n_chunks, n_groups, EU, (long long)ch, skip, dedupe, gsh);
    if (n_supers) VC2_LAUNCH(L, k_index_group<false>, dim3(n_supers, n_pictures), dim3(256), glds, s, lens, payload_stride, (const void *)groups, supers, n_groups, n_supers, EU, (long long)ch * IDX_GROUP, skip, dedupe, gsh);
  }
  // (one lane follows the super-groups of a picture; then a thread per super-group / group expands it: sixteen dependent reads each, all at once)
  VC2_LAUNCH(L, k_index_chain, dim3(n_pictures), dim3(std::min(1024, (n_groups + 63) / 64 * 64)), 0, s, lens, payload_stride, tables, groups, supers, entries, n_chunks, n_groups, n_supers, EU, ch, skip, gsh);
  vc2_prof_end(L, s);
  vc2_prof_begin(L, "slice_index_emit", s);
  VC2_LAUNCH(L, k_index_emit, dim3((n_chunks + 63) / 64, n_pictures), dim3(64), 0, s, payload,
                     payload_stride, lens, entries, offsets, n_chunks, E, n_slices, prefix, scalar, err, ch, skip);
  vc2_prof_end(L, s);
}

// ------------------------------------------------------------------------------------------
// LD slices, decode (legacy profile, cfg 5): one lane per slice, the luma stream then the interleaved
// chroma stream (Slices.cpp:485-560) through the same windowed decoder and staged flush as the HQ slices.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int intlog2_dev(int value) { int l = 0; --value; while (value > 0) { value >>= 1; ++l; } return l; }

__global__ __launch_bounds__(256) void k_ld_unpack(const LdUnpackParams p) {
  constexpr int UNP_N = 32, UNP_PITCH = UNP_N + 4, LPR = UNP_N / 4; // whole 128-byte lines per flush (see k_hq_unpack)
  __shared__ __attribute__((aligned(16))) int stage[4][64 * UNP_PITCH];
  __shared__ unsigned long long outp[4][64];
  __shared__ unsigned short vlut[1024];
  const int pic = blockIdx.y;
  if (p.redo && !p.shifted[pic]) return; // (the second pass: only pictures with a slice whose luma length exceeds the slice)
  vlut_init(vlut);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slice = blockIdx.x * 256 + threadIdx.x;
  const bool chroma = blockIdx.z != 0; // the two streams of a slice are independent once the header is read: a lane each
  const bool active = slice < p.n_slices;
  int *st = stage[wave] + lane * UNP_PITCH;
  const int *sw = stage[wave];
  const uint8_t *data = nullptr;
  int ystart = 0, ybits = 0, total = 0; // bit positions inside the slice
  long long vis = 0;                     // bytes of the payload from the slice's start on that the decoder may look at
  if (active) {
    const int size = p.slice_bytes[slice];
    const unsigned off = p.redo ? p.starts[(size_t)pic * p.n_slices + slice] : p.offsets[slice];
    const long long plen = p.lens ? (long long)min(p.lens[pic], (unsigned long long)p.payload_stride) : p.payload_stride;
    vis = max(plen - (long long)off, 0ll);
    data = p.payload + (size_t)pic * p.payload_stride + min((long long)off, plen);
    unsigned head = 0; // 7 bits of quantiser index, then intlog2(8*size-7) bits of luma length: at most 30 bits
    for (int k = 0; k < 4; ++k) head = (head << 8) | (k < min((long long)(p.redo ? 4 : size), vis) ? data[k] : 0xFFu);
    if (!chroma) p.qidx[(size_t)pic * p.n_slices + slice] = (int)(head >> 25);
    const int split = intlog2_dev(8 * size - 7);
    ybits = split ? (int)((head << 7) >> (32 - split)) : 0;
    ystart = 7 + split;
    total = 8 * size;
    if (!p.redo && p.shifted && !chroma && ybits > total - ystart) atomicOr(&p.shifted[pic], 1u);
    outp[wave][lane] = (unsigned long long)(p.store + (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs);
  } else outp[wave][lane] = 0;
  WordReader br;
  // luma: ybits bits from ystart; bits past the slice read as 1 like bits past the bound (VLC.cpp:182-185).  Second pass:
  // the reference's reader goes on into the bytes behind the slice (bits past the payload read as 1)
  if (!chroma) {
    const int nb = p.redo ? (int)min((long long)ybits, 8 * vis - ystart) : min(ybits, total - ystart);
    if (active && nb > 0) br.init_bits(data + (ystart >> 3), ystart & 7, nb); else br.init_ones();
  }
  const int ny = chroma ? 0 : p.comp_n[0];
  for (int base = 0; base < ny; base += UNP_N) {
    const int room = min(UNP_N, ny - base);
    decode_round<UNP_N, int>(br, room, st, vlut);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < LPR; ++j) {
      const int r = j * (64 / LPR) + lane / LPR, c = (lane % LPR) * 4;
      int32_t *rec = (int32_t *)outp[wave][r];
      if (rec && c < room) {
        const int4 v = *(const int4 *)(sw + r * UNP_PITCH + c);
        int *d = rec + p.comp_off[0] + base + c;
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i vv = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(vv, (__attribute__((address_space(1))) v4i *)(size_t)d);
      }
    }
    wave_lds_sync();
  }
  // chroma: the rest of the slice, U and V coefficients alternating
  if (chroma) {
    const int cstart = ystart + ybits;
    const int nb = p.redo ? (int)min((long long)(total - cstart), 8 * vis - cstart) : total - cstart;
    if (active && nb > 0 && ybits >= 0) br.init_bits(data + (cstart >> 3), cstart & 7, nb); else br.init_ones();
  }
  const int nc = chroma ? 2 * p.comp_n[1] : 0;
  for (int base = 0; base < nc; base += UNP_N) {
    const int room = min(UNP_N, nc - base);
    decode_round<UNP_N, int>(br, room, st, vlut);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < LPR; ++j) {
      const int r = j * (64 / LPR) + lane / LPR, c = (lane % LPR) * 4;
      int32_t *rec = (int32_t *)outp[wave][r];
      if (rec && c < room) { // room is even: (u, v) pairs
        const int4 v = *(const int4 *)(sw + r * UNP_PITCH + c);
        int *du = rec + p.comp_off[1] + (base + c) / 2, *dv = rec + p.comp_off[2] + (base + c) / 2;
        __builtin_nontemporal_store(v.x, du); __builtin_nontemporal_store(v.y, dv);
        if (c + 2 < room) { __builtin_nontemporal_store(v.z, du + 1); __builtin_nontemporal_store(v.w, dv + 1); }
      }
    }
    wave_lds_sync();
  }
}

void vc2_launch_ld_unpack(Launcher &L, const LdUnpackParams &p, int n_pictures, hipStream_t s) {
  vc2_prof_begin(L, "ld_unpack", s);
  VC2_LAUNCH(L, k_ld_unpack, dim3((p.n_slices + 255) / 256, n_pictures, 2), dim3(256), 0, s, p);
  vc2_prof_end(L, s);
}
// The true slice starts of a picture whose flag is up: one lane follows the reference's reader through the slice
// headers (LDSliceIO, Slices.cpp:246-303: 7 bits of index, the luma length, that many bits of luma, the rest of the slice --
// if any is left -- of chroma, then the next byte boundary).  Three dependent loads per slice, for corrupt pictures only.
// A picture that runs past its payload is the reference's "Failed to read LD compressed frame".
__global__ __launch_bounds__(64) void k_ld_walk(const LdUnpackParams p) {
  const int pic = blockIdx.x;
  if (!p.shifted[pic] || threadIdx.x != 0) return;
  const uint8_t *pay = p.payload + (size_t)pic * p.payload_stride;
  const unsigned long long plen = p.lens ? min(p.lens[pic], (unsigned long long)p.payload_stride) : (unsigned long long)p.payload_stride;
  unsigned long long pos = 0;
  for (int si = 0; si < p.n_slices; ++si) {
    const int size = p.slice_bytes[si];
    p.starts[(size_t)pic * p.n_slices + si] = (uint32_t)min(pos, plen);
    unsigned head = 0;
    for (int k = 0; k < 4; ++k) head = (head << 8) | (pos + k < plen ? pay[pos + k] : 0xFFu);
    const int split = intlog2_dev(8 * size - 7);
    const long long ybits = split ? (long long)((head << 7) >> (32 - split)) : 0;
    const long long uvbits = 8ll * size - 7 - split - ybits;
    pos += (unsigned long long)((7 + split + ybits + max(uvbits, 0ll) + 7) >> 3);
  }
  if (pos > plen) atomicOr(p.err, VC2_DEVERR_STREAM);
}
void vc2_launch_ld_walk(Launcher &L, const LdUnpackParams &p, int n_pictures, hipStream_t s) {
  vc2_prof_begin(L, "ld_unpack", s);
  VC2_LAUNCH(L, k_ld_walk, dim3(n_pictures), dim3(64), 0, s, p);
  vc2_prof_end(L, s);
}

// LL band with DC prediction: anti-diagonal wavefront, one workgroup per (picture).
// restored(y,x) = scale(q(y,x), aq(slice of (y,x))) + predictDC(restored, y, x)
__global__ __launch_bounds__(1024) void k_ld_ll(const int32_t *store, long long store_stride,
                                                int slice_coefs, int coef_off, int n0, int llh, int llw,
                                                int ys, int xs, const int32_t *qidx, int qm0,
                                                int32_t *ll_plane, long long ll_stride, unsigned *err) {
  const int pic = blockIdx.x;
  const int32_t *st = store + (size_t)pic * store_stride;
  const int32_t *qi = qidx + (size_t)pic * ys * xs;
  int32_t *ll = ll_plane + (size_t)pic * ll_stride;
  const int bh = llh / ys, bw = llw / xs; // LL block of one slice (bh*bw == n0)
  (void)n0;
  for (int d = 0; d < llh + llw - 1; ++d) {
    const int ylo = max(0, d - (llw - 1)), yhi = min(llh - 1, d);
    for (int y = ylo + (int)threadIdx.x; y <= yhi; y += blockDim.x) {
      const int x = d - y;
      const int sv = y / bh, sh = x / bw;
      const int qv = st[(size_t)(sv * xs + sh) * slice_coefs + coef_off + (y - sv * bh) * bw + (x - sh * bw)];
      // slice whose index quantises this LL sample: Quantisation.cpp:298-299
      const int yb = ((y + 1) * ys - 1) / llh, xb = ((x + 1) * xs - 1) / llw;
      const int aq = max(qi[yb * xs + xb] - qm0, 0);
      if (aq > 119) atomicOr(err, VC2_DEVERR_QINDEX);
      int pred;
      if (y > 0 && x > 0) {
        const int r = ll[(size_t)(y - 1) * llw + x - 1] + ll[(size_t)(y - 1) * llw + x] + ll[(size_t)y * llw + x - 1];
        pred = r >= 0 ? (r + 1) / 3 : (r - 1) / 3;
      } else if (y > 0) pred = ll[(size_t)(y - 1) * llw + x];
      else if (x > 0) pred = ll[(size_t)y * llw + x - 1];
      else pred = 0;
      ll[(size_t)y * llw + x] = scale_dev(qv, min(aq, 119)) + pred;
    }
    __threadfence_block();
    __syncthreads();
  }
}

// DC prediction of a plane of up to 64 R rows in LDS by ONE wavefront, no barrier.  A lane owns R consecutive rows and
// walks the columns one step behind the lane above it; what it needs from there -- the row above its first row, at this
// column and the one before -- arrives by DPP (wave_shr:1) from that lane's last two steps, its own left and upper-left
// neighbours are the previous step's registers, the residuals are fetched a step ahead.  The anti-diagonal sweep over
// 2 x 2 blocks it replaces paid a workgroup barrier and an LDS round trip per diagonal (16 HD pictures: 0.148 ms, of
// which 0.053 are the gather of the residuals and the plane's way out).  A lone wavefront waits ~8 cycles for every
// dependent result and ~20 for every branch, so a step is straight-line code and the chain per sample is four
// instructions (first version, a test per row and the literal formula: 0.138 ms; straight-line: 0.108):
//   * predictDC (Quantisation.cpp:191-208), s >= 0 ? (s + 1) / 3 : (s - 1) / 3 for s = upper-left + up + left, is
//     floor((s + 1) / 3) for either sign; with a bias B = 0x7FFFFFFE (a multiple of 3) the dividend is non-negative as an
//     unsigned word for every s > INT_MIN, and floor(u / 3) = umulhi(u, 0xAAAAAAAB) >> 1 for every 32-bit u.  The bias
//     goes into the upper-left term and B / 3 comes off the residual, both known a step early: add3, mul_hi, shift, add.
//     (s = INT_MIN, where this differs, and s = INT_MAX, where the reference's s + 1 overflows, need |samples| ~ 2^30.)
//   * column 0 (prediction = the sample above) is a prefix sum down the column, done first; row 0 (prediction = the
//     sample to the left) is a select in the lane that owns it.
#ifndef VC2_LD_LL_WAVE_MAXR
#define VC2_LD_LL_WAVE_MAXR 4 // rows per lane up to which a plane goes to ld_ll_wave (0: every plane takes the anti-diagonal sweep)
#endif
template <int R> __device__ __forceinline__ void ld_ll_wave(int *rs, int llh, int llw) {
  constexpr unsigned BIAS = 0x7FFFFFFEu, BIAS3 = BIAS / 3u;
  const int lane = threadIdx.x, y0 = lane * R, nl = (llh + R - 1) / R;
  int row[R]; // LDS offsets of the own rows (rows past the plane: the last row, never stored)
  bool ok[R];
  unsigned left[R], res[R]; // the own rows at column x - 1; their residuals at column x, less BIAS3
  unsigned run = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    ok[r] = y0 + r < llh;
    row[r] = min(y0 + r, llh - 1) * llw;
    run += ok[r] ? (unsigned)rs[row[r]] : 0u;
    left[r] = run; // column 0, still without the rows of the lanes above
  }
  const unsigned above = (unsigned)wave_incl_scan((int)run, lane) - run; // column 0 of the row above the own first row
#pragma unroll
  for (int r = 0; r < R; ++r) {
    left[r] += above;
    if (ok[r]) rs[row[r]] = (int)left[r];
    res[r] = (unsigned)rs[row[r] + min(1, llw - 1)] - BIAS3;
  }
  unsigned bottom = left[R - 1]; // the own last row at the column of the latest step
  unsigned prev_up = above;      // the row above the own first row at column x - 1
  const bool top = y0 == 0;
  for (int t = 0; t < llw - 1 + nl - 1; ++t) {
    const unsigned up_in = (unsigned)dpp0<0x138, 0xf>((int)bottom); // the lane above finished this column in the previous step
    const int x = t - lane + 1;
    if (x >= 1 && x < llw && ok[0]) {
      unsigned nres[R];
      const int xn = min(x + 1, llw - 1);
#pragma unroll
      for (int r = 0; r < R; ++r) nres[r] = (unsigned)rs[row[r] + xn] - BIAS3; // (column x + 1 is the next step's: not yet overwritten)
      unsigned ul = prev_up + (BIAS + 1u), up = up_in;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        unsigned d3 = __umulhi(ul + up + left[r], 0xAAAAAAABu) >> 1;
        if (r == 0) d3 = top ? left[0] + BIAS3 : d3; // row 0 of the plane: the sample to the left (a select; the formula with left standing in for up and upper-left wraps for |left| > 2^31 / 3, as corrupt streams' indices make it)
        const unsigned a = res[r] + d3;
        if (ok[r]) rs[row[r] + x] = (int)a;
        ul = left[r] + (BIAS + 1u);
        left[r] = a;
        up = a;
        res[r] = nres[r];
      }
      bottom = up; // (a partial last lane hands nothing on: no lane below it has rows)
      prev_up = up_in;
    }
  }
}

// The same with the whole LL plane in LDS (it fits for HD pictures: 135 x 240 x 4 bytes): the residuals are dequantised
// into LDS in one parallel sweep, the anti-diagonal wavefront then only touches LDS (a barrier per diagonal costs
// tenths of a microsecond instead of a round trip to memory), and the plane is written out once.
__device__ __forceinline__ void ld_ll_lds_body(int pic, const int32_t *store, long long store_stride,
                                               int slice_coefs, int coef_off, int llh, int llw,
                                               int ys, int xs, const int32_t *qidx, int qm0,
                                               int32_t *ll_plane, long long ll_stride, unsigned *err) {
  extern __shared__ int rs[];
  const int32_t *st = store + (size_t)pic * store_stride;
  const int32_t *qi = qidx + (size_t)pic * ys * xs;
  const int bh = llh / ys, bw = llw / xs; // LL block of one slice
  // quantiser factor / offset by adjusted index behind the plane: a per-lane index into the constant tables is a
  // dependent trip to memory per sample
  int *qft = rs + llh * llw, *qot = qft + 120;
  for (int i = threadIdx.x; i < 120; i += blockDim.x) { qft[i] = c_qs.qf[i]; qot[i] = c_qs.off[i]; }
  __syncthreads();
  // slice by slice: one index per slice (the slice whose index quantises a sample of its own block, Quantisation.cpp:298-299,
  // is the slice itself when the blocks tile the band, which the geometry check guarantees), no divisions per sample
  auto scale_tab = [&](int v, int qf, int qo) -> int { // scale(), Quantisation.cpp:86-95
    const unsigned mag = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    int a = (int)(mag * (unsigned)qf);
    if (a > 0) a = (int)((unsigned)a + (unsigned)qo);
    a = (int)((unsigned)a + 2u);
    a /= 4;
    return v < 0 ? (int)(0u - (unsigned)a) : a;
  };
  const int n0 = bh * bw, ns = ys * xs;
  if (n0 <= 4) { // four slices per thread and turn, their loads in flight together: the records are 2 KB apart, every load a trip to memory
    for (int s0 = threadIdx.x; s0 < ns; s0 += 4 * blockDim.x) {
      int q4[4], v4[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sl = min(s0 + k * (int)blockDim.x, ns - 1);
        q4[k] = qi[sl];
        const int32_t *src = st + (size_t)sl * slice_coefs + coef_off;
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[k][e] = e < n0 ? src[e] : 0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sl = s0 + k * (int)blockDim.x;
        if (sl >= ns) break;
        const int sv = sl / xs, sh = sl - sv * xs;
        const int aq = max(q4[k] - qm0, 0);
        if (aq > 119) atomicOr(err, VC2_DEVERR_QINDEX);
        const int qf = qft[min(aq, 119)], qo = qot[min(aq, 119)];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < n0) { const int yy = e / bw, xx = e - yy * bw; rs[(sv * bh + yy) * llw + sh * bw + xx] = scale_tab(v4[k][e], qf, qo); }
      }
    }
  } else {
    for (int sl = threadIdx.x; sl < ns; sl += blockDim.x) {
      const int sv = sl / xs, sh = sl - sv * xs;
      const int aq = max(qi[sl] - qm0, 0);
      if (aq > 119) atomicOr(err, VC2_DEVERR_QINDEX);
      const int qf = qft[min(aq, 119)], qo = qot[min(aq, 119)];
      const int32_t *src = st + (size_t)sl * slice_coefs + coef_off;
      for (int yy = 0; yy < bh; ++yy)
        for (int xx = 0; xx < bw; ++xx) rs[(sv * bh + yy) * llw + sh * bw + xx] = scale_tab(src[yy * bw + xx], qf, qo);
    }
  }
  __syncthreads();
  // The wavefront over anti-diagonals is a chain of dependent steps, each a barrier and an LDS round trip: it runs over
  // 2 x 2 blocks (half the steps; the four samples of a block follow one another in registers).
  {
    auto pred = [&](int y, int x, int ul, int up, int lf) -> int { // predictDC, Quantisation.cpp:191-208
      if (y > 0 && x > 0) { const int r = ul + up + lf; return r >= 0 ? (r + 1) / 3 : (r - 1) / 3; }
      if (y > 0) return up;
      if (x > 0) return lf;
      return 0;
    };
    // Planes of up to 256 rows (HD: 135): ONE wavefront, no barrier at all.  A lane owns R = ceil(rows / 64) consecutive
    // rows and walks the columns one step behind the lane above it; what it needs from there -- the row above its first
    // row, at this column and the one before -- arrives by DPP (wave_shr:1) from that lane's last two steps, its own
    // left and upper-left neighbours are the previous step's registers.  A step is R dependent predictions (~12
    // instructions each) instead of a workgroup barrier and an LDS round trip per anti-diagonal of blocks:
    // 16 HD pictures 0.148 -> ... ms.
    const int R = (llh + 63) / 64;
    if (R <= VC2_LD_LL_WAVE_MAXR) {
      if (threadIdx.x < 64) {
        if (R == 1) ld_ll_wave<1>(rs, llh, llw);
        else if (R == 2) ld_ll_wave<2>(rs, llh, llw);
        else if (R == 3) ld_ll_wave<3>(rs, llh, llw);
        else ld_ll_wave<4>(rs, llh, llw);
      }
      __syncthreads();
    } else
    for (int d = 0; d < (llh + 1) / 2 + (llw + 1) / 2 - 1; ++d) {
      const int nby = (llh + 1) / 2, nbx = (llw + 1) / 2;
      const int blo = max(0, d - (nbx - 1)), bhi = min(nby - 1, d);
      for (int by = blo + (int)threadIdx.x; by <= bhi; by += blockDim.x) {
        const int y0 = 2 * by, x0 = 2 * (d - by);
        const bool has_x = x0 + 1 < llw, has_y = y0 + 1 < llh;
        int *q = rs + y0 * llw + x0;
        const int w00 = y0 > 0 && x0 > 0 ? q[-llw - 1] : 0, w01 = y0 > 0 ? q[-llw] : 0, w02 = y0 > 0 && has_x ? q[-llw + 1] : 0;
        const int w10 = x0 > 0 ? q[-1] : 0, w20 = x0 > 0 && has_y ? q[llw - 1] : 0;
        const int a = (int)((unsigned)q[0] + (unsigned)pred(y0, x0, w00, w01, w10));
        q[0] = a;
        int b = 0, c = 0;
        if (has_x) { b = (int)((unsigned)q[1] + (unsigned)pred(y0, x0 + 1, w01, w02, a)); q[1] = b; }
        if (has_y) { c = (int)((unsigned)q[llw] + (unsigned)pred(y0 + 1, x0, w10, a, w20)); q[llw] = c; }
        if (has_x && has_y) q[llw + 1] = (int)((unsigned)q[llw + 1] + (unsigned)pred(y0 + 1, x0 + 1, a, b, c));
      }
      __syncthreads();
    }
  }
  int32_t *ll = ll_plane + (size_t)pic * ll_stride;
  for (int i = threadIdx.x; i < llh * llw; i += blockDim.x) ll[i] = rs[i];
}

__global__ __launch_bounds__(1024) void k_ld_ll_lds(const int32_t *store, long long store_stride,
                                                    int slice_coefs, int coef_off, int llh, int llw,
                                                    int ys, int xs, const int32_t *qidx, int qm0,
                                                    int32_t *ll_plane, long long ll_stride, unsigned *err) {
  ld_ll_lds_body(blockIdx.x, store, store_stride, slice_coefs, coef_off, llh, llw, ys, xs, qidx, qm0, ll_plane, ll_stride, err);
}
// the three components of a picture in one launch (grid y): the wavefront is a chain of barriers, not work
__global__ __launch_bounds__(1024) void k_ld_ll_lds3(const LdLl3Params p) {
  const int c = blockIdx.y;
  ld_ll_lds_body(blockIdx.x, p.store, p.store_stride, p.slice_coefs, c == 0 ? p.coef_off[0] : c == 1 ? p.coef_off[1] : p.coef_off[2],
                 c == 0 ? p.llh[0] : c == 1 ? p.llh[1] : p.llh[2], c == 0 ? p.llw[0] : c == 1 ? p.llw[1] : p.llw[2], p.ys, p.xs,
                 p.qidx, p.qm0, c == 0 ? p.ll_plane[0] : c == 1 ? p.ll_plane[1] : p.ll_plane[2],
                 c == 0 ? p.ll_stride[0] : c == 1 ? p.ll_stride[1] : p.ll_stride[2], p.err);
}
bool vc2_launch_ld_ll3(Launcher &L, const LdLl3Params &p, int n_pictures, hipStream_t s) {
  size_t bytes = 0;
  int reach = 0;
  for (int c = 0; c < 3; ++c) {
    if (p.llh[c] < 1 || p.llw[c] < 1) return false;
    bytes = std::max(bytes, (size_t)p.llh[c] * p.llw[c] * 4 + 960); // + factor / offset tables
    reach = std::max(reach, std::min(p.llh[c], p.llw[c]));
  }
  if (bytes > 150 * 1024) return false;
  vc2_prof_begin(L, "ld_ll_predict", s);
  vc2_allow_lds((const void *)k_ld_ll_lds3, 150 * 1024);
  // planes of up to 256 rows: the prediction is one wavefront's work (ld_ll_wave), the other wavefronts only gather the
  // residuals and write the plane out (16 HD pictures: 256 threads 0.094 ms, 512 0.099, 1024 0.100).  Taller planes: 2 x 2
  // blocks on the longest anti-diagonal, never more are busy in a step (measured for 68 blocks: 128 threads 0.167 ms,
  // 256 0.148, 512 0.153: barriers cost more)
  const int blocks = (reach + 1) / 2;
  static const int tune_threads = vc2_tune_int("VC2HIP_LD_LL_THREADS", 0);
  const int threads = tune_threads ? tune_threads : blocks <= 256 ? 256 : 512;
  VC2_LAUNCH(L, k_ld_ll_lds3, dim3(n_pictures, 3), dim3(threads), bytes, s, p);
  vc2_prof_end(L, s);
  return true;
}
void vc2_launch_ld_ll(Launcher &L, const int32_t *store, long long store_stride, int slice_coefs,
                      int coef_off, int n0, int llh, int llw, int ys, int xs, const int32_t *qidx,
                      int qm0, int32_t *ll_plane, long long ll_stride, int n_pictures, unsigned *err,
                      hipStream_t s) {
  vc2_prof_begin(L, "ld_ll_predict", s);
  const size_t plane_bytes = (size_t)llh * llw * 4 + 960; // + factor / offset tables
  if (plane_bytes <= 150 * 1024) {
    vc2_allow_lds((const void *)k_ld_ll_lds, 150 * 1024);
    VC2_LAUNCH(L, k_ld_ll_lds, dim3(n_pictures), dim3(1024), plane_bytes, s, store, store_stride, slice_coefs, coef_off,
                       llh, llw, ys, xs, qidx, qm0, ll_plane, ll_stride, err);
  } else {
    VC2_LAUNCH(L, k_ld_ll, dim3(n_pictures), dim3(1024), 0, s, store, store_stride, slice_coefs, coef_off, n0,
                       llh, llw, ys, xs, qidx, qm0, ll_plane, ll_stride, err);
  }
  vc2_prof_end(L, s);
}

// ------------------------------------------------------------------------------------------
// LD encode (legacy profile): per-slice quantiser search with the DC-prediction state machine
// (EncodeStream.cpp:141-245 quantIndicesLD + SliceQuantiserRef), DC-predicted quantisation of the
// LL band (Quantisation.cpp:213-234) and the LD slice writer (Slices.cpp:195-244).
//
// A slice's LL predictions read the reconstructed LL samples of the slices above, left and
// above-left only, so the slices of one anti-diagonal are independent: one launch per diagonal,
// one wavefront per slice.  The last pass of a slice (at its chosen index) leaves the quantised
// coefficients -- LL band as prediction residuals -- in the coefficient store, in coding order.
// ------------------------------------------------------------------------------------------
// GLOBAL: a slice too large for LDS (the reference's only limit is sliceSizeIsValid: slices up to the whole picture) -- the
// coefficients are read from the store in every trial and the trial's quantised values go to a scratch array of the
// store's shape; the lanes of the wavefront hand them to each other through memory (workgroup-scope fences: one CU).
template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_ld_quantise_diag(const LdEncParams p, int d) {
  extern __shared__ __attribute__((aligned(16))) int lds_i[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pic = blockIdx.y;
  const int sv = max(0, d - (p.xs - 1)) + (int)blockIdx.x * (int)(blockDim.x >> 6) + wave;
  if (sv > min(p.ys - 1, d)) return; // no workgroup barriers below
  const int sh = d - sv, slice = sv * p.xs + sh;
  int32_t *rec = p.store + (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs;
  const int staged = GLOBAL ? 0 : 2 * p.slice_coefs; // LDS ints per wavefront for the two copies
  int *co = GLOBAL ? rec : lds_i + wave * staged;
  int *qv = GLOBAL ? p.scratch + (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs : co + p.slice_coefs;
  auto wave_sync = [&]() {
    if constexpr (GLOBAL) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else wave_lds_sync();
  };
#define wave_lds_sync wave_sync
  if constexpr (!GLOBAL) for (int i = lane; i < p.slice_coefs; i += 64) co[i] = rec[i];
  wave_lds_sync();

  // quantise the whole slice at index tq into qv; true if an adjusted index leaves the table
  const int wpw = blockDim.x >> 6;
  uint4 *qtab = (uint4 *)(lds_i + wpw * staged) + wave * 32; // per subband: magic, shift, factor
  const int n_bands = 3 * p.depth + 1;
  // reconstructed LL samples of the slice's blocks with one row above and one column to the left (the neighbours'
  // final values, loaded once): the trials then run without touching memory
  int *rs = lds_i + wpw * (staged + 32 * 4) + wave * p.rs_ints;
  {
    int *r0 = rs;
    for (int c = 0; c < 3; ++c) {
      if (!p.comp_n[c]) continue;
      const int32_t *res = p.restored[c] + (size_t)pic * p.restored_stride[c];
      const int llw = p.ll_w[c], bh = p.bh[c], bw = p.bw[c], pitch = bw + 1;
      for (int i = lane; i < bh + bw + 1; i += 64) {
        const int yy = i <= bw ? 0 : i - bw, xx = i <= bw ? i : 0; // row 0: bw + 1 samples, then column 0
        const int y = sv * bh - 1 + yy, x = sh * bw - 1 + xx;
        r0[yy * pitch + xx] = (y >= 0 && x >= 0) ? res[(size_t)y * llw + x] : 0;
      }
      r0 += (bh + 1) * pitch;
    }
  }
  auto quantise_at = [&](int tq) -> bool {
    bool bad = false;
    wave_lds_sync();
    if (lane < n_bands) {
      const int aq = max(tq - p.qmatrix[lane], 0);
      if (aq > 119) bad = true;
      const int a = min(aq, 119);
      qtab[lane] = make_uint4(c_qs.magic[a], (unsigned)c_qs.shift[a], (unsigned)c_qs.qf[a], 0u);
    }
    wave_lds_sync();
    if (lane == 0) { // LL blocks: serial raster scan, prediction from the reconstructed samples (rs: block + halo, LDS)
      const int aq = max(tq - p.qmatrix[0], 0);
      if (aq > 119) bad = true;
      int *r0 = rs;
      for (int c = 0; c < 3 && !bad; ++c) {
        if (!p.comp_n[c]) continue;
        const int bh = p.bh[c], bw = p.bw[c], pitch = bw + 1;
        for (int yy = 0; yy < bh; ++yy)
          for (int xx = 0; xx < bw; ++xx) {
            const int y = sv * bh + yy, x = sh * bw + xx;
            const int *up = r0 + yy * pitch + xx; // up-left; up + 1: above; up + pitch: left
            int pred; // predictDC, Quantisation.cpp:191-208
            if (y > 0 && x > 0) {
              const int r = up[0] + up[1] + up[pitch];
              pred = r >= 0 ? (r + 1) / 3 : (r - 1) / 3;
            } else if (y > 0) pred = up[1];
            else if (x > 0) pred = up[pitch];
            else pred = 0;
            const int v = co[p.comp_off[c] + yy * bw + xx];
            const int qq = quant_dev((int)((unsigned)v - (unsigned)pred), aq);
            r0[(yy + 1) * pitch + xx + 1] = (int)((unsigned)scale_dev(qq, aq) + (unsigned)pred);
            qv[p.comp_off[c] + yy * bw + xx] = qq;
          }
        r0 += (bh + 1) * pitch;
      }
    }
    for (int c = 0; c < 3; ++c) {
      const int n = p.comp_n[c], n0 = p.comp_n0[c], off = p.comp_off[c];
      if (!n) continue;
      const int n0_shift = (n0 & (n0 - 1)) == 0 ? 31 - __clz(n0) : -1;
      for (int j = n0 + lane; j < n; j += 64) {
        const uint4 t = qtab[band_of_index_fast(j, n0, n0_shift)];
        qv[off + j] = quant_core(co[off + j], (int)t.z, t.x, (int)t.y);
      }
    }
    wave_lds_sync(); // qv is complete before any lane measures it
    return __any(bad);
  };
  // luma_slice_bits + chroma_slice_bits, Slices.cpp:51-95 (U and V coefficients alternate)
  auto need_bits = [&]() -> int {
    auto ly = [&](int j) -> int { return qv[p.comp_off[0] + j]; };
    auto lc = [&](int j) -> int { return qv[p.comp_off[1 + (j & 1)] + (j >> 1)]; };
    return component_bits<false, false>(ly, p.comp_n[0], 1, 0, nullptr, lane, nullptr, 0, p.err) +
           component_bits<false, false>(lc, 2 * p.comp_n[1], 1, 0, nullptr, lane, nullptr, 0, p.err);
  };

  int q;
  bool bad = false;
  if (p.search) {
    const int bytes = p.slice_bytes[slice];
    const int avail = 8 * bytes - 7 - intlog2_dev(8 * bytes - 7);
    int trial = 63, delta = 64;
    q = 127;
    while (delta > 0) {
      delta >>= 1;
      if ((bad = quantise_at(trial))) break;
      if (need_bits() <= avail) { if (trial < q) q = trial; trial -= delta; }
      else trial += delta;
    }
    if (lane == 0) p.qidx[(size_t)pic * p.n_slices + slice] = q;
  } else {
    q = p.qidx[(size_t)pic * p.n_slices + slice];
  }
  if (!bad) bad = quantise_at(q);
  if (bad) { if (lane == 0) atomicOr(p.err, VC2_DEVERR_QINDEX); return; }
  wave_lds_sync();
  for (int i = lane; i < p.slice_coefs; i += 64) rec[i] = qv[i];
  { // the slice's reconstructed LL samples become the neighbours' halo
    const int *r0 = rs;
    for (int c = 0; c < 3; ++c) {
      if (!p.comp_n[c]) continue;
      int32_t *res = p.restored[c] + (size_t)pic * p.restored_stride[c];
      const int llw = p.ll_w[c], bh = p.bh[c], bw = p.bw[c], pitch = bw + 1;
      for (int i = lane; i < bh * bw; i += 64) {
        const int yy = i / bw, xx = i - yy * bw;
        res[(size_t)(sv * bh + yy) * llw + sh * bw + xx] = r0[(yy + 1) * pitch + xx + 1];
      }
      r0 += (bh + 1) * pitch;
    }
  }
#undef wave_lds_sync
}

// Tables of the fast anti-diagonal step, built once per batch: subband of every luma / interleaved chroma index
// (2 x 512 bytes), the 120 quantiser entries (magic, shift, factor, offset), the quantisation matrix (32 ints)
constexpr int LD_TAB_INTS = 256 + 480 + 32;
__global__ __launch_bounds__(256) void k_ld_tables(const LdEncParams p) {
  unsigned char *band_y = (unsigned char *)p.tab, *band_uv = band_y + 512;
  uint4 *qs = (uint4 *)(p.tab + 256);
  int *qm = p.tab + 736;
  const int n_y = p.comp_n[0], n_c = p.comp_n[1], n0y = p.comp_n0[0], n0c = p.comp_n0[1];
  const int sy = (n0y & (n0y - 1)) == 0 ? 31 - __clz(n0y) : -1, sc = (n0c & (n0c - 1)) == 0 ? 31 - __clz(n0c) : -1;
  for (int j = threadIdx.x; j < 512; j += blockDim.x) {
    band_y[j] = (unsigned char)band_of_index_fast(min(j, n_y - 1), n0y, sy);
    band_uv[j] = (unsigned char)band_of_index_fast(min(j >> 1, n_c - 1), n0c, sc);
  }
  for (int j = threadIdx.x; j < 120; j += blockDim.x)
    qs[j] = make_uint4(c_qs.magic[j], (unsigned)c_qs.shift[j], (unsigned)c_qs.qf[j], (unsigned)c_qs.off[j]);
  for (int j = threadIdx.x; j < 32; j += blockDim.x) qm[j] = j < 3 * p.depth + 1 ? p.qmatrix[j] : 0;
}

// The same anti-diagonal step for the common geometry (a component of at most 512 / 256 coefficients, its LL block
// inside the first eight).  The chain of anti-diagonals is latency bound, so what counts is the length of the
// dependent chain inside a step.  One workgroup of four wavefronts per slice; every wavefront keeps the slice's
// coefficients (CPL per lane, luma / interleaved chroma stream) and their matrix entries in registers.  The seven
// bisection steps of quantIndicesLD (EncodeStream.cpp:141-190) are taken in four rounds of 2 + 2 + 2 + 1 steps: a
// round measures the three indices its two steps can visit (the trial, the trial -/+ half the step).  In a round
// wavefront 0 runs the DC-predicted LL chains of the candidates side by side (one lane per candidate and component,
// each on its own copy of the reconstructed LL block), wavefronts 1-3 measure the remaining subbands of one
// candidate each; the walk over the results is the reference's loop.  Ten measurements instead of seven, a
// dependent chain of four instead of seven.  All global loads of a step are issued together, nothing is written
// until the index is final.
constexpr int LD_ABORT_AT = -1; // LDS word in front of llv (the last word of the tables' padding): the row was given up
constexpr int LD_SLOTS = 16; // candidates of round r: 4 r + (0: trial, 1: trial - step, 2: trial + step); 15: an index that was no trial

// ROWS: one workgroup per row of slices of a picture walks the row (d = pictures in the batch, see k_ld_search_rows)
template <int CPL, bool DUAL, bool ROWS>
__device__ __forceinline__ void ld_diag_body(const LdEncParams &p, int d, int rs_pad) {
  extern __shared__ __attribute__((aligned(16))) int lds_i[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // ROWS: workgroup = row * P8 + picture with P8 = pictures rounded up to the 8 XCDs.  Workgroups go round-robin to the
  // XCDs and each XCD starts its share in order, so all rows of a picture run on one XCD (the hand-over between rows
  // stays in that XCD's L2: measured 3.2 ms against 3.7 ms with rows scattered) and the row above always started first.
  const int p8 = (d + 7) & ~7;
  const int pic = ROWS ? (int)blockIdx.x % p8 : (int)blockIdx.y;
  if (ROWS && pic >= d) return;
  const int n_y = p.comp_n[0], n_c = p.comp_n[1], n_uv = 2 * n_c, n0y = p.comp_n0[0], n0c = p.comp_n0[1];
  const unsigned char *band_y = (const unsigned char *)lds_i, *band_uv = band_y + 512;
  const uint4 *qs = (const uint4 *)(lds_i + 256);
  const int *qmt = lds_i + 736;
  int *llv = lds_i + LD_TAB_INTS;          // LL coefficients: [0,8) luma, [8,16) U/V alternating
  int *rsS = llv + 16;                     // per candidate: LL blocks with the row above and the column to the left
  int *llq = rsS + LD_SLOTS * rs_pad;      // per candidate: quantised LL residuals, laid out like llv
  int *llpack = llq + LD_SLOTS * 16;       // per candidate and component: code lengths of the residuals, a byte each
  int *acres = llpack + LD_SLOTS * 8;      // per candidate: luma count, chroma count (behind the LL blocks), bad index
  const int sv = ROWS ? (int)blockIdx.x / p8 : max(0, d - (p.xs - 1)) + (int)blockIdx.x;
#ifdef VC2HIP_ABLATE // tests/test_gpu_parity.py::test_ld_handoff_failure_path: one row of slices of picture 0 never comes
  if (ROWS && p.debug_dead_row >= 0 && sv == p.debug_dead_row && pic == 0) return;
#endif
  const int sh_first = ROWS ? 0 : d - sv, sh_end = ROWS ? p.xs : sh_first + 1;
  constexpr bool dual = DUAL; // n_y, n_uv <= 32 CPL: luma on lanes 0-31, chroma on lanes 32-63, one pass per candidate
  const bool ch = dual && lane >= 32;
  // slot A: luma (dual: chroma on the upper lanes); slot B: chroma when not dual
  const int j0 = (dual ? lane & 31 : lane) * CPL;
  const int nA = ch ? n_uv : n_y, nB = dual ? 0 : n_uv;
  const int llA = ch ? 2 * n0c : n0y, llB = 2 * n0c; // stream indices below these are LL
  for (int i = threadIdx.x * 4; i < LD_TAB_INTS; i += blockDim.x * 4) *(int4 *)(lds_i + i) = *(const int4 *)(p.tab + i);
  __syncthreads();
  if (threadIdx.x == 0) llv[LD_ABORT_AT] = 0;
  int qmA[CPL], qmB[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    qmA[k] = qmt[(ch ? band_uv : band_y)[min(j0 + k, 511)]];
    qmB[k] = qmt[band_uv[min(j0 + k, 511)]];
  }
  const int qm0 = qmt[0];
  for (int sh = sh_first; sh < sh_end; ++sh) {
  const int slice = sv * p.xs + sh;
  int32_t *rec = p.store + (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs;
  int vA[CPL], vB[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) vA[k] = vB[k] = 0;
  auto load_y = [&](int (&v)[CPL]) {
    if (j0 < n_y) { const int4 a = *(const int4 *)(rec + p.comp_off[0] + j0); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
    if constexpr (CPL == 8)
      if (j0 + 4 < n_y) { const int4 a = *(const int4 *)(rec + p.comp_off[0] + j0 + 4); v[4] = a.x; v[5] = a.y; v[6] = a.z; v[7] = a.w; }
  };
  auto load_uv = [&](int (&v)[CPL]) {
    if (j0 < n_uv) {
      if constexpr (CPL == 8) {
        const int4 a = *(const int4 *)(rec + p.comp_off[1] + j0 / 2), b = *(const int4 *)(rec + p.comp_off[2] + j0 / 2);
        v[0] = a.x; v[1] = b.x; v[2] = a.y; v[3] = b.y; v[4] = a.z; v[5] = b.z; v[6] = a.w; v[7] = b.w;
      } else {
        const int2 a = *(const int2 *)(rec + p.comp_off[1] + j0 / 2), b = *(const int2 *)(rec + p.comp_off[2] + j0 / 2);
        v[0] = a.x; v[1] = b.x; v[2] = a.y; v[3] = b.y;
      }
    }
  };
  if (ch) load_uv(vA); else load_y(vA);
  if (!dual) load_uv(vB);
  int bytes = 0, qfix = 0;
  if (p.search) bytes = p.slice_bytes[slice]; else qfix = p.qidx[(size_t)pic * p.n_slices + slice];
  if (wave == 0) { // reconstructed LL samples around the slice's blocks (row above: bw + 1 samples, then the column to the left), into every candidate's copy
    if (ROWS && sv > 0) { // the slice above (and with it the one above-left) is final once its index is published
      // Hand-over between workgroups of one launch, in the form of cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md
      // "Valid forms" that needs no cache maintenance: EVERY store of the handed-over samples and of the flag is a
      // device-coherent write-through store (sc1: relaxed agent-scope atomics), the storing wavefront drains them
      // (s_waitcnt vmcnt(0)) before it stores the flag, and EVERY load of them here is an sc1 load issued by the
      // wavefront that polled, after its poll matched.  Nothing depends on dispatch order or placement for
      // correctness: a workgroup that waits for one that never comes gives up (bounded wait), poisons its own row so
      // the rows below give up at once, and reports VC2_DEVERR_HANDOFF; the library then repeats nothing silently --
      // it reports the error and uses the per-diagonal launches from then on (vc2_launch_ld_quantise).
      const int32_t *flag = p.qidx + (size_t)pic * p.n_slices + (size_t)(sv - 1) * p.xs + sh;
      int spins = 0, f;
      while ((f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == -1) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 18)) { f = -2; break; }
      }
      if (f == -2) { // the row above is dead (or never came): stop this row too
        if (lane == 0) {
          atomicOr(p.err, VC2_DEVERR_HANDOFF);
          for (int s2 = sh; s2 < p.xs; ++s2)
            __hip_atomic_store(p.qidx + (size_t)pic * p.n_slices + (size_t)sv * p.xs + s2, -2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        llv[LD_ABORT_AT] = 1;
      }
    }
    const int h0 = p.bh[0] + p.bw[0] + 1, h1 = p.bh[1] + p.bw[1] + 1, h2 = p.bh[2] + p.bw[2] + 1;
    const int o1 = (p.bh[0] + 1) * (p.bw[0] + 1), o2 = o1 + (p.bh[1] + 1) * (p.bw[1] + 1);
    for (int e = lane; e < h0 + h1 + h2; e += 64) { // one load per lane, all components in flight together
      const int c = e < h0 ? 0 : e < h0 + h1 ? 1 : 2, i = e - (c == 0 ? 0 : c == 1 ? h0 : h0 + h1);
      const int32_t *res = (c == 0 ? p.restored[0] : c == 1 ? p.restored[1] : p.restored[2]) +
                           (size_t)pic * (c == 0 ? p.restored_stride[0] : c == 1 ? p.restored_stride[1] : p.restored_stride[2]);
      const int llw = c == 0 ? p.ll_w[0] : c == 1 ? p.ll_w[1] : p.ll_w[2];
      const int bh = c == 0 ? p.bh[0] : c == 1 ? p.bh[1] : p.bh[2], bw = c == 0 ? p.bw[0] : c == 1 ? p.bw[1] : p.bw[2];
      const int yy = i <= bw ? 0 : i - bw, xx = i <= bw ? i : 0;
      const int y = sv * bh - 1 + yy, x = sh * bw - 1 + xx;
      int val = 0;
      if (y >= 0 && x >= 0) // ROWS: written by other workgroups of this launch -- device-coherent accesses, no cache maintenance
        val = ROWS ? __hip_atomic_load(res + (size_t)y * llw + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : res[(size_t)y * llw + x];
      const int at = (c == 0 ? 0 : c == 1 ? o1 : o2) + yy * (bw + 1) + xx;
      for (int sl = 0; sl < LD_SLOTS; ++sl) rsS[sl * rs_pad + at] = val;
    }
    if (j0 < 8) { // the LL coefficients go to the chains
#pragma unroll
      for (int k = 0; k < CPL; ++k) llv[(ch ? 8 : 0) + j0 + k] = vA[k];
      if (!dual) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) llv[8 + j0 + k] = vB[k];
      }
    }
  }
  __syncthreads();
  if (ROWS && llv[LD_ABORT_AT]) return; // every wavefront of the workgroup: the row was given up (see above)

  // component c's LL chain of candidate slot sl at index tq: raster scan, prediction from the reconstructed samples
  auto chain = [&](int sl, int c, int tq) {
    if (tq - qm0 > 119) return; // reported by the subband pass
    const int bh = c == 0 ? p.bh[0] : c == 1 ? p.bh[1] : p.bh[2], bw = c == 0 ? p.bw[0] : c == 1 ? p.bw[1] : p.bw[2];
    int *r0 = rsS + sl * rs_pad + (c == 0 ? 0 : (p.bh[0] + 1) * (p.bw[0] + 1) + (c == 1 ? 0 : (p.bh[1] + 1) * (p.bw[1] + 1)));
    const int ll0 = c == 0 ? 0 : 7 + c, step = c == 0 ? 1 : 2;
    const uint4 t = qs[max(tq - qm0, 0)];
    const int pitch = bw + 1;
    int pack0 = 0, pack1 = 0, i = 0;
    for (int yy = 0; yy < bh; ++yy)
      for (int xx = 0; xx < bw; ++xx, ++i) {
        const int y = sv * bh + yy, x = sh * bw + xx;
        const int *up = r0 + yy * pitch + xx; // up-left; up + 1: above; up + pitch: left
        int pred; // predictDC, Quantisation.cpp:191-208
        if (y > 0 && x > 0) {
          const int r = up[0] + up[1] + up[pitch];
          pred = r >= 0 ? (r + 1) / 3 : (r - 1) / 3;
        } else if (y > 0) pred = up[1];
        else if (x > 0) pred = up[pitch];
        else pred = 0;
        int qq = quant_core((int)((unsigned)llv[ll0 + i * step] - (unsigned)pred), (int)t.z, t.x, (int)t.y);
        const int nb = svlc_bits(qq); // (beyond 32 bits: measured like any other, see component_bits; the slice writer raises the error if such an index is chosen)
        // scale(), Quantisation.cpp:86-95, with the table's factor and offset
        const unsigned mag = qq < 0 ? 0u - (unsigned)qq : (unsigned)qq;
        int r = (int)(mag * t.z);
        if (r > 0) r = (int)((unsigned)r + t.w);
        r = (int)((unsigned)r + 2u);
        r /= 4;
        if (qq < 0) r = (int)(0u - (unsigned)r);
        r0[(yy + 1) * pitch + xx + 1] = (int)((unsigned)r + (unsigned)pred);
        llq[sl * 16 + ll0 + i * step] = qq;
        if (i < 4) pack0 |= nb << (8 * i); else pack1 |= nb << (8 * (i - 4));
      }
    llpack[sl * 8 + 2 * c] = pack0;
    llpack[sl * 8 + 2 * c + 1] = pack1;
  };
  // one stream at index tq: bits of the lane's coefficients behind the LL block (sum) and the end of its last non-zero code
  auto eval = [&](const int (&v)[CPL], const int (&qm)[CPL], int n, int n_ll, int tq, bool &over, int &sum, int &le) {
    sum = 0; le = 0;
    if (j0 >= n) return;
    unsigned a8[CPL], qf8[CPL], dom = 0;
    int qq[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int aq = max(tq - qm[k], 0);
      over = over || aq > 119;
      const uint4 t = qs[min(aq, 119)];
      const unsigned a = (v[k] < 0 ? 0u - (unsigned)v[k] : (unsigned)v[k]) << 2;
      const unsigned m = __umulhi(t.x, a);
      qq[k] = (int)((m + ((a - m) >> 1)) >> t.y);
      a8[k] = a; qf8[k] = t.z;
      dom |= a | (t.z - 2u);
    }
    if (__any((int)dom < 0)) { // outside the reciprocal's domain: the literal division (see quant_core)
#pragma unroll
      for (int k = 0; k < CPL; ++k)
        if ((int)(a8[k] | (qf8[k] - 2u)) < 0) qq[k] = (int)a8[k] / (int)qf8[k];
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k) { // only the length matters: the sign of the quantised value does not change it
      const unsigned m1 = (unsigned)(qq[k] < 0 ? -qq[k] : qq[k]) + 1u;
      const int nb = qq[k] == 0 ? 1 : 2 * (31 - __clz((int)m1)) + 2; // (beyond 32 bits: a length like any other, see component_bits)
      if (j0 + k < n && j0 + k >= n_ll) {
        sum += nb;
        if (qq[k] != 0) le = sum;
      }
    }
  };
  // the subbands of candidate slot sl at index tq: coded bits of each stream behind its LL block, up to the last
  // non-zero coefficient
  auto ac = [&](int sl, int tq) {
    bool over = tq - qm0 > 119;
    int sumA, leA, sumB = 0, leB = 0;
    eval(vA, qmA, nA, llA, tq, over, sumA, leA);
    if (!dual) eval(vB, qmB, nB, llB, tq, over, sumB, leB);
    const bool any_over = __any(over);
    int cy, cc;
    int incl = wave_incl_scan(sumA, lane);
    if (dual) {
      const int total_y = __builtin_amdgcn_readlane(incl, 31);
      int m = leA ? incl - sumA - (ch ? total_y : 0) + leA : 0; // prefix maximum inside each half
      m = max(m, dpp0<0x111, 0xf>(m));
      m = max(m, dpp0<0x112, 0xf>(m));
      m = max(m, dpp0<0x114, 0xf>(m));
      m = max(m, dpp0<0x118, 0xf>(m));
      m = max(m, dpp0<0x142, 0xa>(m));
      cy = __builtin_amdgcn_readlane(m, 31);
      cc = __builtin_amdgcn_readlane(m, 63);
    } else {
      cy = wave_max(leA ? incl - sumA + leA : 0);
      incl = wave_incl_scan(sumB, lane);
      cc = wave_max(leB ? incl - sumB + leB : 0);
    }
    if (lane == 0) { acres[sl * 4] = cy; acres[sl * 4 + 1] = cc; acres[sl * 4 + 2] = any_over ? 1 : 0; }
  };
  auto run_round = [&](int slot0, int ncand, int trial, int step) { // candidates: trial, trial - step, trial + step
    auto tq_of = [&](int cand) { return cand == 0 ? trial : cand == 1 ? trial - step : trial + step; };
    if (wave == 0) {
      const int cand = lane >> 2, c = lane & 3;
      if (cand < ncand && c < 3) chain(slot0 + cand, c, tq_of(cand));
    } else {
      for (int cand = wave - 1; cand < ncand; cand += nw - 1) ac(slot0 + cand, tq_of(cand));
    }
    __syncthreads();
  };
  // coded bits of the slice for candidate slot sl (luma_slice_bits + chroma_slice_bits, Slices.cpp:51-95: each stream up
  // to its last non-zero coefficient)
  auto need_of = [&](int sl) -> int {
    const int cy = acres[sl * 4], cc = acres[sl * 4 + 1];
    const unsigned y0 = (unsigned)llpack[sl * 8], y1 = (unsigned)llpack[sl * 8 + 1];
    const unsigned u0 = (unsigned)llpack[sl * 8 + 2], v0 = (unsigned)llpack[sl * 8 + 4];
    int ytot = 0, ylast = 0, ctot = 0, clast = 0; // LL sections: total bits, end of the last non-zero code (>= 4 bits)
    for (int i = 0; i < n0y; ++i) {
      const int nb = (int)(((i < 4 ? y0 : y1) >> (8 * (i & 3))) & 0xFFu);
      ytot += nb;
      if (nb > 1) ylast = ytot;
    }
    for (int i = 0; i < n0c; ++i) {
      const int nu = (int)((u0 >> (8 * i)) & 0xFFu), nv = (int)((v0 >> (8 * i)) & 0xFFu);
      ctot += nu;
      if (nu > 1) clast = ctot;
      ctot += nv;
      if (nv > 1) clast = ctot;
    }
    return (cy ? ytot + cy : ylast) + (cc ? ctot + cc : clast);
  };

  int q, slot = LD_SLOTS - 1;
  bool bad = false;
  if (p.search) {
    const int avail = 8 * bytes - 7 - intlog2_dev(8 * bytes - 7);
    int trial = 63, delta = 64;
    q = 127;
    for (int round = 0; round < 4 && !bad; ++round) {
      const int steps = round < 3 ? 2 : 1, ncand = round < 3 ? 3 : 1, slot0 = 4 * round;
      run_round(slot0, ncand, trial, delta >> 1);
      // lane c measures candidate c; the walk is the reference's loop on the two ballots
      const bool mine = lane < ncand;
      const unsigned long long is_bad = __ballot(mine && acres[(slot0 + (mine ? lane : 0)) * 4 + 2] != 0);
      const unsigned long long fits = __ballot(mine && need_of(slot0 + (mine ? lane : 0)) <= avail);
      int cand = 0;
      for (int k = 0; k < steps; ++k) {
        delta >>= 1;
        if ((is_bad >> cand) & 1) { bad = true; break; }
        if ((fits >> cand) & 1) { if (trial < q) { q = trial; slot = slot0 + cand; } trial -= delta; cand = 1; }
        else { trial += delta; cand = 2; }
      }
    }
    if (!ROWS && threadIdx.x == 0) p.qidx[(size_t)pic * p.n_slices + slice] = q;
  } else {
    q = qfix;
  }
  if (!bad && (q == 127 || !p.search)) { // not a trial of the search: its LL chains now
    slot = LD_SLOTS - 1;
    run_round(slot, 1, q, 0);
    if (acres[slot * 4 + 2]) bad = true;
  }
  if (bad) {
    if (threadIdx.x == 0) atomicOr(p.err, VC2_DEVERR_QINDEX);
    if (!ROWS) return;
    q = 0; // the rows below still get their flag
  }
  if (bad) {
  } else if (wave == 1) { // the quantised slice: LL residuals from the chains, everything else through the table
    auto quant8 = [&](const int (&v)[CPL], const int (&qm)[CPL], int n_ll, const int *ll, int (&out)[CPL]) {
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const uint4 t = qs[min(max(q - qm[k], 0), 119)];
        const int r = quant_core(v[k], (int)t.z, t.x, (int)t.y);
        out[k] = j0 + k < n_ll ? ll[min(j0 + k, 7)] : r;
      }
    };
    auto store_y = [&](const int (&o)[CPL]) {
      if (j0 < n_y) *(int4 *)(rec + p.comp_off[0] + j0) = make_int4(o[0], o[1], o[2], o[3]);
      if constexpr (CPL == 8)
        if (j0 + 4 < n_y) *(int4 *)(rec + p.comp_off[0] + j0 + 4) = make_int4(o[4], o[5], o[6], o[7]);
    };
    auto store_uv = [&](const int (&o)[CPL]) {
      if (j0 < n_uv) {
        if constexpr (CPL == 8) {
          *(int4 *)(rec + p.comp_off[1] + j0 / 2) = make_int4(o[0], o[2], o[4], o[6]);
          *(int4 *)(rec + p.comp_off[2] + j0 / 2) = make_int4(o[1], o[3], o[5], o[7]);
        } else {
          *(int2 *)(rec + p.comp_off[1] + j0 / 2) = make_int2(o[0], o[2]);
          *(int2 *)(rec + p.comp_off[2] + j0 / 2) = make_int2(o[1], o[3]);
        }
      }
    };
    int out[CPL];
    if (j0 < nA) {
      quant8(vA, qmA, llA, llq + slot * 16 + (ch ? 8 : 0), out);
      if (ch) store_uv(out); else store_y(out);
    }
    if (!dual && j0 < nB) {
      quant8(vB, qmB, llB, llq + slot * 16 + 8, out);
      store_uv(out);
    }
  } else if (wave == 0) { // the slice's reconstructed LL samples become the neighbours' halo
    const int *r0 = rsS + slot * rs_pad;
    for (int c = 0; c < 3; ++c) {
      int32_t *res = p.restored[c] + (size_t)pic * p.restored_stride[c];
      const int llw = p.ll_w[c], bh = p.bh[c], bw = p.bw[c], pitch = bw + 1;
      for (int i = lane; i < bh * bw; i += 64) {
        const int yy = i / bw, xx = i - yy * bw;
        int32_t *at = res + (size_t)(sv * bh + yy) * llw + sh * bw + xx;
        if (ROWS) __hip_atomic_store(at, r0[(yy + 1) * pitch + xx + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *at = r0[(yy + 1) * pitch + xx + 1];
      }
      r0 += (bh + 1) * pitch;
    }
  }
  if (ROWS && wave == 0) { // publish: the reconstructed samples first, then the index as the row's progress flag
    // this wavefront made every one of the write-through stores above: drain them, then the flag (inline asm: the
    // compiler may not drop or move this wait -- MI355X_MICROARCH.md "Compiler hazard")
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(p.qidx + (size_t)pic * p.n_slices + slice, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  } // next slice of the row
}

template <int CPL, bool DUAL>
__global__ __launch_bounds__(256) void k_ld_quantise_diag_fast(const LdEncParams p, int d, int rs_pad) { ld_diag_body<CPL, DUAL, false>(p, d, rs_pad); }
// The whole search in one launch: workgroup (row sv, picture) walks its row of slices left to right; slice (sv, sh) starts
// when row sv - 1 has published slice sh (p.qidx doubles as the progress flag: -1 until the slice is final).  A workgroup
// only ever waits for one that was dispatched before it; the wait is bounded all the same (error flag, no hang).
template <int CPL, bool DUAL>
__global__ __launch_bounds__(256) void k_ld_search_rows(const LdEncParams p, int n_pictures, int rs_pad) { ld_diag_body<CPL, DUAL, true>(p, n_pictures, rs_pad); }
// LD slice writer: one wavefront per slice, image assembled in LDS as big-endian words
__global__ __launch_bounds__(256) void k_ld_pack(const LdEncParams p) {
  extern __shared__ unsigned lds_u[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slice = blockIdx.x * (blockDim.x >> 6) + wave, pic = blockIdx.y;
  if (slice >= p.n_slices) return;
  const int size = p.slice_bytes[slice];
  unsigned *img = lds_u + wave * p.img_words;
  for (int i = lane; i < p.img_words; i += 64) img[i] = 0;
  wave_lds_sync();
  const int32_t *rec = p.store + (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs;
  auto ly = [&](int j) -> int { return rec[p.comp_off[0] + j]; };
  auto lc = [&](int j) -> int { return rec[p.comp_off[1 + (j & 1)] + (j >> 1)]; };
  const int split = intlog2_dev(8 * size - 7);
  const int n_y = p.comp_n[0], n_uv = 2 * p.comp_n[1];
  if (n_y <= 512 && n_uv <= 512) { // one round per stream: codes and lengths once, in registers
    // codes + lengths of eight values of a stream, computed (LD magnitudes are small, a table would cost more to load)
    auto codes = [&](Coef8 &c, const int (&raw)[8], int j0, int n) {
      c.sum = 0; c.last_end = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int v = raw[k], nb = svlc_bits(v);
        unsigned code = svlc_code(v);
        if (nb > 32) { atomicOr(p.err, VC2_DEVERR_CODE32); nb = 1; code = 1; v = 0; }
        c.nb[k] = j0 + k < n ? nb : 0;
        c.code[k] = code;
        c.sum += c.nb[k];
        if (v != 0 && c.nb[k]) c.last_end = c.sum;
      }
    };
    const int j0 = lane * 8;
    int raw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = 0;
    if (j0 < n_y) { const int4 a = *(const int4 *)(rec + p.comp_off[0] + j0); raw[0] = a.x; raw[1] = a.y; raw[2] = a.z; raw[3] = a.w; }
    if (j0 + 4 < n_y) { const int4 a = *(const int4 *)(rec + p.comp_off[0] + j0 + 4); raw[4] = a.x; raw[5] = a.y; raw[6] = a.z; raw[7] = a.w; }
    Coef8 cy, cc;
    codes(cy, raw, j0, n_y);
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = 0;
    if (j0 < n_uv) { // U and V alternate (n_uv is a multiple of 8)
      const int4 a = *(const int4 *)(rec + p.comp_off[1] + j0 / 2), b = *(const int4 *)(rec + p.comp_off[2] + j0 / 2);
      raw[0] = a.x; raw[1] = b.x; raw[2] = a.y; raw[3] = b.y; raw[4] = a.z; raw[5] = b.z; raw[6] = a.w; raw[7] = b.w;
    }
    codes(cc, raw, j0, n_uv);
    const int iy = wave_incl_scan(cy.sum, lane), ic = wave_incl_scan(cc.sum, lane);
    const int ybits = wave_max(cy.last_end ? iy - cy.sum + cy.last_end : 0);
    const int cbits = wave_max(cc.last_end ? ic - cc.sum + cc.last_end : 0);
    const int uvbits = 8 * size - 7 - split - ybits;
    if (uvbits < cbits) { if (lane == 0) atomicOr(p.err, VC2_DEVERR_LD_TOOBIG); return; }
    if (lane == 0) {
      put_code(img, 0, (unsigned)p.qidx[(size_t)pic * p.n_slices + slice] & 127u, 7);
      if (split > 0) put_code(img, 7, (unsigned)ybits, split);
    }
    // bounded writes: codes wholly past the bound are 1-bits of trailing zeros and are dropped
    // (VLC.cpp:151-172); flush pads the bound with 0 bits, which the zeroed image already holds
    write8(img, 7 + split + iy - cy.sum, 7 + split + ybits, cy);
    write8(img, 7 + split + ybits + ic - cc.sum, 7 + split + ybits + uvbits, cc);
    wave_lds_sync();
    uint8_t *out = p.payload + (size_t)pic * p.payload_stride + p.offsets[slice];
    for (int k = lane; k < size; k += 64) out[k] = (uint8_t)(img[k >> 2] >> (24 - 8 * (k & 3)));
    return;
  }
  const int ybits = component_bits<false, false>(ly, p.comp_n[0], 1, 0, nullptr, lane, nullptr, 0, p.err);
  const int cbits = component_bits<false, false>(lc, 2 * p.comp_n[1], 1, 0, nullptr, lane, nullptr, 0, p.err);
  const int uvbits = 8 * size - 7 - split - ybits;
  if (uvbits < cbits) { if (lane == 0) atomicOr(p.err, VC2_DEVERR_LD_TOOBIG); return; }
  if (lane == 0) {
    put_code(img, 0, (unsigned)p.qidx[(size_t)pic * p.n_slices + slice] & 127u, 7);
    if (split > 0) put_code(img, 7, (unsigned)ybits, split);
  }
  // bounded writes: codes wholly past the bound are 1-bits of trailing zeros and are dropped
  // (VLC.cpp:151-172); flush pads the bound with 0 bits, which the zeroed image already holds
  component_bits<false, true>(ly, p.comp_n[0], 1, 0, nullptr, lane, img, ybits, p.err, 7 + split);
  component_bits<false, true>(lc, 2 * p.comp_n[1], 1, 0, nullptr, lane, img, uvbits, p.err, 7 + split + ybits);
  wave_lds_sync();
  uint8_t *out = p.payload + (size_t)pic * p.payload_stride + p.offsets[slice];
  for (int k = lane; k < size; k += 64) out[k] = (uint8_t)(img[k >> 2] >> (24 - 8 * (k & 3)));
}

// set once a hand-over of the single-launch search timed out (vc2hip_sync saw VC2_DEVERR_HANDOFF): from then on the
// search runs as one launch per anti-diagonal of slices, which needs no hand-over inside a launch
static std::atomic<bool> g_ld_rows_disabled{false}; // (set by any context's error path, read by every launcher thread)
bool vc2_ld_rows_disabled() { return g_ld_rows_disabled; }
void vc2_ld_disable_rows() { g_ld_rows_disabled = true; }

void vc2_launch_ld_quantise(Launcher &L, const LdEncParams &p, int n_pictures, hipStream_t s) {
  vc2_prof_begin(L, p.search ? "ld_search" : "ld_quantise", s);
  const bool fast = p.comp_n[0] <= 512 && p.comp_n[1] <= 256 && p.comp_n[1] == p.comp_n[2] && p.comp_n[1] > 0 && 3 * p.depth + 1 <= 32 &&
                    p.comp_n0[0] <= 8 && p.comp_n0[1] <= 4 && p.comp_n[0] % 4 == 0 && p.comp_off[0] == 0 &&
                    p.comp_off[1] == p.comp_n[0] && p.comp_off[2] == p.comp_n[0] + p.comp_n[1];
  if (fast) {
    const int rs_pad = p.rs_ints | 1; // LL blocks + halo of one candidate (odd: the candidates' copies in different banks)
    const size_t lds = (size_t)(LD_TAB_INTS + 16 + LD_SLOTS * (rs_pad + 16 + 8 + 4)) * 4;
    const bool small = p.comp_n[0] <= 256 && 2 * p.comp_n[1] <= 256; // four coefficients per lane cover a stream
    const int reach = small ? 128 : 256;
    const bool dual = p.comp_n[0] <= reach && 2 * p.comp_n[1] <= reach; // half a wavefront covers a stream
    VC2_LAUNCH(L, k_ld_tables, dim3(1), dim3(256), 0, s, p);
    const int rows = !p.diagonals; // (VC2HIP_FLAG_LD_DIAGONALS)
#ifdef VC2HIP_ABLATE
    LdEncParams pd = p;
    { const char *e = getenv("VC2HIP_DEBUG_LD_DEAD_ROW"); pd.debug_dead_row = e ? atoi(e) : -1; }
    const LdEncParams &p = pd; // (shadows the argument for the launches below)
#endif
    if (rows && p.search && !vc2_ld_rows_disabled()) {
      // 3 wavefronts (LL chains + 2 x subbands) measured fastest: 3.2 ms per 16 HD pictures, 4 wavefronts 3.6, one launch per anti-diagonal 3.7-3.9
      static const int nwr = std::min(4, std::max(2, vc2_tune_int("VC2HIP_LD_WAVES", 3)));
      (void)hipMemsetAsync(p.qidx, 0xFF, (size_t)n_pictures * p.n_slices * sizeof(int32_t), s); // progress flags: -1 = not final
      const dim3 grid(p.ys * ((n_pictures + 7) & ~7)), blk(64 * nwr);
      if (small && dual) VC2_LAUNCH(L, (k_ld_search_rows<4, true>), grid, blk, lds, s, p, n_pictures, rs_pad);
      else if (small) VC2_LAUNCH(L, (k_ld_search_rows<4, false>), grid, blk, lds, s, p, n_pictures, rs_pad);
      else if (dual) VC2_LAUNCH(L, (k_ld_search_rows<8, true>), grid, blk, lds, s, p, n_pictures, rs_pad);
      else VC2_LAUNCH(L, (k_ld_search_rows<8, false>), grid, blk, lds, s, p, n_pictures, rs_pad);
      vc2_prof_end(L, s);
      return;
    }
    for (int d = 0; d < p.ys + p.xs - 1; ++d) {
      const int cnt = min(p.ys - 1, d) - max(0, d - (p.xs - 1)) + 1;
      const dim3 grid(cnt, n_pictures);
      static const int nwv = std::min(4, std::max(2, vc2_tune_int("VC2HIP_LD_WAVES", 4)));
      const dim3 blk(64 * nwv);
      if (small && dual) VC2_LAUNCH(L, (k_ld_quantise_diag_fast<4, true>), grid, blk, lds, s, p, d, rs_pad); // 75 registers: forcing 64 for full residency of a 16-picture diagonal spills and measured slower
      else if (small) VC2_LAUNCH(L, (k_ld_quantise_diag_fast<4, false>), grid, blk, lds, s, p, d, rs_pad);
      else if (dual) VC2_LAUNCH(L, (k_ld_quantise_diag_fast<8, true>), grid, blk, lds, s, p, d, rs_pad);
      else VC2_LAUNCH(L, (k_ld_quantise_diag_fast<8, false>), grid, blk, lds, s, p, d, rs_pad);
    }
    vc2_prof_end(L, s);
    return;
  }
  size_t per_wave = (size_t)2 * p.slice_coefs * 4 + 32 * 16 + (size_t)p.rs_ints * 4; // coefficients, quantised copy, subband table, LL blocks + halo
  const bool global = per_wave > 160 * 1024 && p.scratch; // the slice stays in the store (the launcher's caller provided the scratch array)
  if (global) per_wave = 32 * 16 + (size_t)p.rs_ints * 4;
  const int wpw = std::max(1, vc2_waves_for_lds(per_wave));
  vc2_allow_lds((const void *)k_ld_quantise_diag<false>, 160 * 1024);
  vc2_allow_lds((const void *)k_ld_quantise_diag<true>, 160 * 1024);
  for (int d = 0; d < p.ys + p.xs - 1; ++d) {
    const int cnt = min(p.ys - 1, d) - max(0, d - (p.xs - 1)) + 1;
    if (global) VC2_LAUNCH(L, k_ld_quantise_diag<true>, dim3((cnt + wpw - 1) / wpw, n_pictures), dim3(64 * wpw), wpw * per_wave, s, p, d);
    else VC2_LAUNCH(L, k_ld_quantise_diag<false>, dim3((cnt + wpw - 1) / wpw, n_pictures), dim3(64 * wpw), wpw * per_wave, s, p, d);
  }
  vc2_prof_end(L, s);
}
void vc2_launch_ld_pack(Launcher &L, const LdEncParams &p, int n_pictures, hipStream_t s) {
  const size_t per_wave = (size_t)p.img_words * 4;
  const int wpw = vc2_waves_for_lds(per_wave);
  vc2_allow_lds((const void *)k_ld_pack, 160 * 1024);
  vc2_prof_begin(L, "ld_pack", s);
  VC2_LAUNCH(L, k_ld_pack, dim3((p.n_slices + wpw - 1) / wpw, n_pictures), dim3(64 * wpw), wpw * per_wave, s, p);
  vc2_prof_end(L, s);
}

// ------------------------------------------------------------------------------------------
// layout conversion + stand-alone (de)quantisation for the fine-grained API
// ------------------------------------------------------------------------------------------
// index of in-place plane position (yy,xx) of a slice tile inside the component record
__device__ __forceinline__ int record_index(int yy, int xx, int sh, int sw, int depth, int *band_out) {
  const int D = depth;
  const int a = yy ? __ffs(yy) - 1 : 31, b = xx ? __ffs(xx) - 1 : 31;
  const int m = min(a, b);
  const int n0 = (sh >> D) * (sw >> D);
  if (m >= D) { *band_out = 0; return (yy >> D) * (sw >> D) + (xx >> D); }
  const int Lv = D - m, s = 1 << (m + 1);
  const int kind = (a > m) ? 0 : (b > m ? 1 : 2);
  const int band = 3 * (Lv - 1) + 1 + kind;
  *band_out = band;
  return band_offset(n0, band) + (yy / s) * (sw / s) + (xx / s);
}

__global__ void k_plane_to_store(const int32_t *plane, int ph, int pw, int depth, int ys, int xs,
                                 int32_t *store, int slice_coefs, int coef_off) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)ph * pw) return;
  const int y = (int)(i / pw), x = (int)(i % pw);
  const int sh = ph / ys, sw = pw / xs;
  const int sv = y / sh, shh = x / sw;
  int band;
  const int idx = record_index(y - sv * sh, x - shh * sw, sh, sw, depth, &band);
  store[(size_t)(sv * xs + shh) * slice_coefs + coef_off + idx] = plane[i];
}

__global__ void k_store_to_plane(const int32_t *store, int slice_coefs, int coef_off, int32_t *plane,
                                 int ph, int pw, int depth, int ys, int xs, const int32_t *qidx,
                                 const int *qmatrix, int mode, unsigned *err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)ph * pw) return;
  const int y = (int)(i / pw), x = (int)(i % pw);
  const int sh = ph / ys, sw = pw / xs;
  const int sv = y / sh, shh = x / sw;
  int band;
  const int idx = record_index(y - sv * sh, x - shh * sw, sh, sw, depth, &band);
  int v = store[(size_t)(sv * xs + shh) * slice_coefs + coef_off + idx];
  if (mode == 1) {
    const int aq = max(qidx[sv * xs + shh] - qmatrix[band], 0);
    if (aq > 119) atomicOr(err, VC2_DEVERR_QINDEX);
    v = scale_dev(v, min(aq, 119));
  }
  plane[i] = v;
}

__global__ void k_quantise_store(int32_t *store, int n_slices, int slice_coefs, int comp_n, int comp_off,
                                 int n0, const int32_t *qidx, const int *qmatrix, unsigned *err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_slices * comp_n) return;
  const int slice = (int)(i / comp_n), j = (int)(i % comp_n);
  const int aq = max(qidx[slice] - qmatrix[band_of_index(j, n0)], 0);
  int32_t *p = &store[(size_t)slice * slice_coefs + comp_off + j];
  if (aq > 119) { atomicOr(err, VC2_DEVERR_QINDEX); return; }
  *p = quant_dev(*p, aq);
}

void vc2_launch_plane_to_store(Launcher &L, const int32_t *plane, int ph, int pw, int depth, int ys,
                               int xs, int32_t *store, int slice_coefs, int coef_off, hipStream_t s) {
  const size_t n = (size_t)ph * pw;
  vc2_prof_begin(L, "plane_to_store", s);
  VC2_LAUNCH(L, k_plane_to_store, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, plane, ph, pw, depth, ys, xs,
                     store, slice_coefs, coef_off);
  vc2_prof_end(L, s);
}
void vc2_launch_store_to_plane(Launcher &L, const int32_t *store, int slice_coefs, int coef_off,
                               int32_t *plane, int ph, int pw, int depth, int ys, int xs,
                               const int32_t *qidx, const int *qmatrix, int mode, unsigned *err,
                               hipStream_t s) {
  const size_t n = (size_t)ph * pw;
  vc2_prof_begin(L, "store_to_plane", s);
  VC2_LAUNCH(L, k_store_to_plane, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, store, slice_coefs, coef_off,
                     plane, ph, pw, depth, ys, xs, qidx, qmatrix, mode, err);
  vc2_prof_end(L, s);
}
void vc2_launch_quantise_store(Launcher &L, int32_t *store, int n_slices, int slice_coefs, int comp_n,
                               int comp_off, int n0, const int32_t *qidx, const int *qmatrix,
                               unsigned *err, hipStream_t s) {
  const size_t n = (size_t)n_slices * comp_n;
  vc2_prof_begin(L, "quantise_store", s);
  VC2_LAUNCH(L, k_quantise_store, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, store, n_slices, slice_coefs,
                     comp_n, comp_off, n0, qidx, qmatrix, err);
  vc2_prof_end(L, s);
}

__global__ void k_fill_i32(int32_t *p, int32_t v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_fill_u64(unsigned long long *p, unsigned long long v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void vc2_launch_fill_i32(Launcher &L, int32_t *p, int32_t v, size_t n, hipStream_t s) {
  vc2_prof_begin(L, "fill", s);
  VC2_LAUNCH(L, k_fill_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
  vc2_prof_end(L, s);
}
void vc2_launch_fill_u64(Launcher &L, unsigned long long *p, unsigned long long v, size_t n, hipStream_t s) {
  vc2_prof_begin(L, "fill", s);
  VC2_LAUNCH(L, k_fill_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
  vc2_prof_end(L, s);
}
