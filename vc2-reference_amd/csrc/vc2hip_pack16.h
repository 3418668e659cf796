// k_hq_pack16 -- the HQ slice coder for the common slice geometry on the 16-bit coefficient store: ONE round per slice,
// sixteen coefficients per lane.  (Included by vc2hip_slices.hip behind its scalar helpers; same reference functions:
// HQSliceIO_VBR / _CBR Slices.cpp:305-382, :469-533, component_slice_bytes :97-119, quant Quantisation.cpp:69-76,
// SignedVLC VLC.cpp:21-52, bounded write / flush VLC.cpp:151-213.)
//
// Why another kernel: k_hq_pack spends ~47 vector instructions per coefficient and lane (two rounds of eight coefficients,
// per-coefficient subband look-ups, a scan and a set of reductions per round, eight codes and eight lengths alive per
// round) and is bound by instruction issue, not by its 1.02 x algorithmic traffic (profiles/r03_*).  Here
//   * a component record is split by the host into a HEAD -- the leading subbands whose blocks are not multiples of sixteen
//     coefficients (LL and the deepest levels: 32 luma / 16 chroma coefficients for 32 x 16 slices at depth 4) -- coded one
//     coefficient per lane with the code computed arithmetically (any magnitude up to the reference's 32-bit code limit),
//     and a BODY of whole sixteen-coefficient runs that lie inside one subband each: a lane's sixteen coefficients share one
//     quantiser factor (no per-coefficient subband look-up), luma on lanes 0-31, U on 32-47, V on 48-63;
//   * quantise = one float multiply and one truncating conversion of the SIGNED value (exact: |v| < 2^15, see
//     k_cbr_search_reg's proof for |v| < 2^20), the low byte of the quotient indexes a 256-entry table of
//     (length << 24 | code with its sign bit) -- address, code and length without shifts or sign handling;
//   * codes are merged pair-wise in 32-bit registers (two codes of <= 14 bits), pairs into quads and quads into ONE string of
//     <= 64 bits per eight coefficients; only the two strings, their lengths and the end of the last non-zero code survive;
//   * head and body bit counts share one DPP scan (packed 16 + 16 bits), the component lengths come from a ballot and a
//     v_readlane of the last lane that has a non-zero coefficient (positions grow with the lane number), so every
//     per-slice quantity is a scalar;
//   * anything outside that domain -- a body quotient beyond +-126, a string of more than 64 bits per eight coefficients, an
//     escape of the 16-bit store -- sends the WHOLE slice (wave-uniform) through component_bits, the general two-pass
//     coder of this file: correct for every input, about three times slower for that slice.
// The launcher (vc2_launch_pack) chooses this kernel when pack16_plan() accepts the geometry; every other geometry, the
// int32 store, unquantised input and images beyond LDS keep k_hq_pack.  Round 6: MODE 2 of the kernel packs VBR pictures in
// ONE pass (look-back over the tiles' byte counts: see the comment in front of the kernel).
#pragma once

#ifdef VC2HIP_ABLATE // why wavefronts leave the table path (tools/probe: VC2HIP_P16_STATS=1 prints the counters after every launch)
__device__ unsigned g_p16_stats[8];
#define P16_STAT(k, cond) do { if (VC2_SKIP(p, 8) && __any(cond) && lane == 0) atomicAdd(&g_p16_stats[k], 1u); } while (0)
#else
#define P16_STAT(k, cond)
#endif
constexpr int P16_LUT_N = 256;            // quotient as a signed byte: entries 0..127 = +0..+127, 128..255 = -128..-1
constexpr int P16_MAXQ = 126;             // |quotient| the table path takes: codes of <= 14 bits, pairs of <= 28
__device__ unsigned g_vlc_lut_s[P16_LUT_N]; // length << 24 | (non-zero ? 0xFF : 0) << 16 | code with its sign bit (<= 14 bits)

static void fill_vlc_lut_s(unsigned *host) {
  for (int i = 0; i < P16_LUT_N; ++i) {
    const int t = i < 128 ? i : i - 256;
    const unsigned m = (unsigned)(t < 0 ? -t : t);
    unsigned code = 1, nb = 1;
    if (m) {
      const unsigned v = m + 1;
      int k = 31;
      while (!((v >> k) & 1u)) --k;
      code = 0;
      for (int b = k - 1; b >= 0; --b) code = (code << 2) | ((v >> b) & 1u);
      code = (((code << 1) | 1u) << 1) | (t < 0 ? 1u : 0u); // (0 b)* 1 s
      nb = 2 * (unsigned)k + 2;
    }
    host[i] = (nb << 24) | (m ? 0xFF0000u : 0u) | code; // every field a byte or a word of its own: SDWA operand selects
  }
}

// by quantiser index: (magic, shift, factor) of the exact integer division and the rounded-up 4 / factor as a float -- the
// rows of c_qs the coder needs, side by side: one 16-byte load per thread instead of four; [120..127]: a factor of 2^30 and
// a reciprocal of zero (indices beyond the table quantise to zero)
__device__ uint4 g_p16_qt[128];
void vc2_upload_p16_tables(const QuantTables &t, hipStream_t s) {
  static uint4 host[128]; // (contexts are created concurrently: filled once, every context only copies it; the values are the same for all)
  static std::once_flag once;
  std::call_once(once, [&t] {
    for (int q = 0; q < 128; ++q) {
      if (q < 120) { union { float f; unsigned u; } w; w.f = t.inv4[q]; host[q] = make_uint4(t.magic[q], (unsigned)t.shift[q], (unsigned)t.qf[q], w.u); }
      else host[q] = make_uint4(0u, 0u, 0x40000000u, 0u);
    }
  });
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_p16_qt), host, sizeof host, 0, hipMemcpyHostToDevice, s);
}
static void vc2_upload_vlc_lut_s(hipStream_t s) {
  static unsigned host[P16_LUT_N];
  static std::once_flag once;
  std::call_once(once, [] { fill_vlc_lut_s(host); });
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_vlc_lut_s), host, sizeof host, 0, hipMemcpyHostToDevice, s);
}

// ---- host: which lanes take what -----------------------------------------------------------------------------------
// Component c owns lanes [lo, lo + width) = [0, 32), [32, 48), [48, 64).  Its first head[c] coefficients (the subbands
// whose blocks are not whole runs of sixteen) go one per lane, the rest sixteen per lane, both from lane lo upwards:
//   lane16[lane]   = quantisation-matrix entry of the lane's body run | of its head coefficient << 8
//   lane16[64 + c] = head[c],  lane16[67 + c] = body lanes of component c
// (the offsets of a lane's coefficients follow from those six numbers by arithmetic: no load in front of the record loads)
#ifndef VC2_P16LB_WAVES
#define VC2_P16LB_WAVES 4 // slices (= wavefronts) of a look-back tile
#endif
constexpr int P16LB_WAVES = VC2_P16LB_WAVES;
static_assert(P16LB_WAVES >= 4 && P16LB_WAVES <= 16 && (P16LB_WAVES & (P16LB_WAVES - 1)) == 0, "the leader of a tile is tile & (waves - 1); the tables are filled by 256 threads");
static bool pack16_plan(const PackParams &p, unsigned *lane16) {
  if (!p.store16 || !p.quantise || p.tile_slices) return false;
  if (p.lookback && (p.cbr_bytes || (p.n_slices + P16LB_WAVES - 1) / P16LB_WAVES > 65535)) return false; // (one pass: VBR only; the tiles are the grid's y)
  if ((p.slice_coefs & 7) || (p.store_stride & 7)) return false; // every record on a 16-byte boundary (the uint4 loads)
  const int lo[3] = {0, 32, 48}, width[3] = {32, 16, 16};
  for (int l = 0; l < 128; ++l) lane16[l] = 0;
  int body_lanes = 0;
  for (int c = 0; c < 3; ++c) {
    const int n = p.comp_n[c], n0 = p.comp_n0[c];
    if (n <= 0 || n0 <= 0 || (p.comp_off[c] & 7)) return false;
    // subband b covers [start, start + size): LL, then three bands per level from the deepest, each level four times larger
    int start = 0, head = -1;
    for (int b = 0; b < 3 * p.depth + 1; ++b) {
      const int size = b == 0 ? n0 : n0 << (2 * ((b - 1) / 3));
      if (p.qmatrix[b] < 0 || p.qmatrix[b] > 255) return false;
      if (head < 0 && (size & 15) == 0 && (start & 15) == 0) head = start; // this band and all behind it: whole runs of 16
      if (head < 0) {
        if (start + size > width[c]) return false; // more head coefficients than the component has lanes
        for (int j = start; j < start + size; ++j) lane16[lo[c] + j] |= (unsigned)p.qmatrix[b] << 8;
      } else {
        if ((start + size - head) / 16 > width[c]) return false;
        for (int j = start; j < start + size; j += 16) lane16[lo[c] + (j - head) / 16] |= (unsigned)p.qmatrix[b];
      }
      start += size;
    }
    if (start != n) return false;
    if (head < 0) head = n;
    if (head & 7) return false; // the body's 16-byte loads
    lane16[64 + c] = (unsigned)head;
    lane16[67 + c] = (unsigned)((n - head) / 16);
    body_lanes += (n - head) / 16;
  }
  return body_lanes >= 40; // fewer: the slices are small, k_hq_pack puts two or four of them on a wavefront
}

// ---- device ----------------------------------------------------------------------------------------------------------
// eight coefficients (four dwords of 16-bit pairs) -> one string: G (right aligned), its length L, end of the last non-zero code.
// Every field of a table entry is a byte or a word of its own, so that lengths, codes and the non-zero mask are operand
// selects (SDWA) of the adds, shifts and ORs that use them; `last` is a running maximum of (bits so far AND mask) -- no
// compare, no select, nothing through VCC.
__device__ __forceinline__ void p16_group(const uint4 w, const float f, const unsigned *lut, unsigned long long &G, int &L,
                                          int &last, float &maxf) {
  const unsigned ww[4] = {w.x, w.y, w.z, w.w};
  unsigned pr[4];
  int pl[4];
  int S = 0;
  last = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float f0 = (float)(int)(short)(ww[d] & 0xFFFFu), f1 = (float)((int)ww[d] >> 16);
    maxf = fmaxf(maxf, fmaxf(__builtin_fabsf(f0), __builtin_fabsf(f1)));
    const int t0 = (int)(f0 * f), t1 = (int)(f1 * f); // quant(): truncation towards zero of the signed quotient
    const unsigned e0 = lut[t0 & 0xFF], e1 = lut[t1 & 0xFF];
    const int l0 = (int)(e0 >> 24), l1 = (int)(e1 >> 24);
    S += l0;
    const int c0 = S & (int)(signed char)(e0 >> 16);
    S += l1;
    const int c1 = S & (int)(signed char)(e1 >> 16);
    last = max(last, max(c0, c1));
    pr[d] = ((e0 & 0xFFFFu) << l1) | (e1 & 0xFFFFu);
    pl[d] = l0 + l1;
  }
  const unsigned long long q0 = ((unsigned long long)pr[0] << pl[1]) | pr[1], q1 = ((unsigned long long)pr[2] << pl[3]) | pr[3];
  G = (q0 << (pl[2] + pl[3])) | q1; // (meaningless beyond 64 bits: the caller tests L)
  L = S;
}


// OR the first `keep` of the n <= 63 right-aligned bits of G into the image at bit position pos (0 <= keep <= n; what a
// bounded write drops beyond the component's length are the '1's of trailing zeros, VLC.cpp:151-156).  No special cases
// and no branch: with keep == 0 the value is zero and the three words it is OR-ed into may lie up to ~70 bytes behind the
// wavefront's image (a run wholly beyond the component's room) -- the next wavefront's image or, for the last one, the
// guard the launcher allocates behind the images (P16_GUARD_BYTES).
__device__ __forceinline__ void p16_put(unsigned *img, int pos, unsigned long long G, int n, int keep) {
  const unsigned long long v = (G >> (n - keep)) << ((64 - keep) & 63); // left aligned (keep == 0: G >> n is zero, the count 0)
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  unsigned *at = img + (pos >> 5);
  const unsigned bo = (unsigned)pos; // (v_alignbit takes the low five bits)
  atomicOr(at, __builtin_amdgcn_alignbit(0u, hi, bo));
  atomicOr(at + 1, __builtin_amdgcn_alignbit(hi, lo, bo));
  atomicOr(at + 2, __builtin_amdgcn_alignbit(lo, 0u, bo));
}

// One component of a slice the table path does not take, a coefficient per lane and trip (rolled: this path must not cost
// the kernel registers): measure (WRITE false: returns component_slice_bytes' bit count) or write the codes that fit.
template <bool WRITE>
__device__ __forceinline__ int p16_general(const int16_t *src, const int32_t *srcw, int n, int n0, int q, const int *qm, int lane,
                                        unsigned *img, int bit0, int region_bits, unsigned *err) {
  const int n0_shift = (n0 & (n0 - 1)) == 0 ? 31 - __clz(n0) : -1;
  int base = 0, count = 0;
#pragma unroll 1
  for (int r0 = 0; r0 < n; r0 += 64) {
    const int j = r0 + lane;
    int c = 0, nb = 0;
    if (j < n) {
      const int aq = max(q - qm[band_of_index_fast(j, n0, n0_shift)], 0);
      if (aq > 119) atomicOr(err, VC2_DEVERR_QINDEX); // (and the coefficient counts as zero, as in k_hq_pack)
      else c = quant_dev(St<int16_t>::load1(src + j, srcw + j), aq);
      nb = svlc_bits(c);
      if (nb > 32) { atomicOr(err, VC2_DEVERR_CODE32); c = 0; nb = 1; } // VLC.h:27: no code beyond 32 bits
    }
    const int incl = wave_incl_scan(nb, lane), pos = base + incl - nb;
    if (WRITE) { if (nb && pos + nb <= region_bits) put_code(img, bit0 + pos, svlc_code(c), nb); }
    else count = max(count, wave_max(c != 0 ? pos + nb : 0));
    base += __builtin_amdgcn_readlane(incl, 63);
  }
  return count;
}

// MODE 0: VBR through slots (scan + compaction behind the kernel), 1: HQ_CBR (every slice at its budget's offset),
// 2: VBR in ONE pass (round 6) -- the slice offsets by decoupled look-back over the workgroups' byte counts, every slice
// written where it belongs in the payload: no slots, no scan, no compaction (1.25 GB written and 2.5 GB moved again per 128
// UHD pictures).  One 8-byte status word per tile of four slices: flag << 62 | bytes (flag 1: the tile's own count, 2: the
// count of everything up to and including it).  The grid of mode 2 is (pictures, tiles): consecutive workgroups are the
// SAME tile of consecutive pictures, so a picture's tile t - 1 started a whole row of workgroups before tile t and the
// chains of all pictures advance side by side (round 1's form in k_hq_pack -- tiles numbered by a ticket per picture, one
// picture's 4050 tiles in flight at a time -- took twice the time of the slots).  Workgroups start in the order of their
// linear index (x fastest), so a predecessor is running or done: no ticket.
constexpr int P16_SLOTS = 0, P16_CBR = 1, P16_LOOKBACK = 2;
#ifndef VC2_P16LB_WPE
#define VC2_P16LB_WPE 8 // mode 2: wavefronts per SIMD the register allocator is held to (64 registers and 12 bytes of spill; left alone it takes 66: 1.72 against 1.69 ms)
#endif
template <int MODE>
__global__ __launch_bounds__((MODE == 2 ? 64 * P16LB_WAVES : 256), (MODE == 2 ? VC2_P16LB_WPE : 1)) void k_hq_pack16(const PackParams p) {
  constexpr bool CBR = MODE == P16_CBR, LB = MODE == P16_LOOKBACK;
  extern __shared__ unsigned lds_u[];
  constexpr int NW = MODE == 2 ? P16LB_WAVES : 4; // slices of a workgroup
  __shared__ int s_tot[NW];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pic = LB ? blockIdx.x : blockIdx.y, tile = LB ? blockIdx.y : blockIdx.x;
  const int slice = tile * NW + wave;
  const bool active = slice < p.n_slices; // wave-uniform
  const int img_q = ((p.prefix + 4 + 3 * 255 * p.scalar + 3) / 4 + 2 + 3) / 4; // image size in 16-byte pieces
  unsigned *lut = lds_u;                   // at LDS address 0: a look-up's address is the quotient's low byte times four
  uint4 *qt = (uint4 *)(lds_u + P16_LUT_N); // by quantiser index: g_p16_qt
  unsigned *img = lds_u + P16_LUT_N + 4 * 128 + wave * img_q * 4;
  // the tables' loads first (L2), the record's behind them: the table writes and the barrier then wait for the former only
  unsigned lut_e = 0x04000002u;
  uint4 qt_e = make_uint4(0u, 0u, 0x40000000u, 0u);
  if (!VC2_SKIP(p, 4)) { // (ablation 4: the tables without their loads -- what amortising the set-up over several slices could gain: 6 %)
    lut_e = g_vlc_lut_s[threadIdx.x & (P16_LUT_N - 1)];
    qt_e = g_p16_qt[threadIdx.x & 127];
  }
  const unsigned lt = p.lane16[lane];
  const int comp = lane < 32 ? 0 : (lane < 48 ? 1 : 2), cl = lane - (lane < 32 ? 0 : (lane < 48 ? 32 : 48));
  const int head_n = (int)(comp == 0 ? p.lane16[64] : (comp == 1 ? p.lane16[65] : p.lane16[66]));
  const int body_n = (int)(comp == 0 ? p.lane16[67] : (comp == 1 ? p.lane16[68] : p.lane16[69]));
  const int coff = comp == 0 ? p.comp_off[0] : (comp == 1 ? p.comp_off[1] : p.comp_off[2]);
  const bool has_body = cl < body_n, has_head = cl < head_n;
  const size_t rec_at = (size_t)pic * p.store_stride + (size_t)(active ? slice : 0) * p.slice_coefs;
  const int16_t *rec = (const int16_t *)p.store + rec_at;
  uint4 w0 = make_uint4(0u, 0u, 0u, 0u), w1 = w0;
  int hv = 0;
  if (active && has_body) { const int16_t *b = rec + coff + head_n + 16 * cl; w0 = *(const uint4 *)b; w1 = *(const uint4 *)(b + 8); }
  if (active && has_head) hv = rec[coff + cl];
  const int32_t *hwide = p.store_wide + rec_at + coff + cl; // the head coefficient's place in the wide array (an escape of the 16-bit store)
  const int q = p.qidx[(size_t)pic * p.n_slices + (active ? slice : 0)]; // (a scalar load: requested before the barrier, used behind it)
  if (threadIdx.x < P16_LUT_N) lut[threadIdx.x] = lut_e;
  if (threadIdx.x < 128) qt[threadIdx.x] = qt_e;
  for (int i = lane; i < img_q; i += 64) ((uint4 *)img)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  if (!active) {
    if constexpr (LB) { // (a picture's last tile: the wavefronts without a slice count as zero bytes and keep the barriers' company)
      if (lane == 0) s_tot[wave] = 0;
      __syncthreads();
      __syncthreads();
    }
    return;
  }

  const int aqb = max(q - (int)(lt & 0xFFu), 0), aqh = max(q - (int)((lt >> 8) & 0xFFu), 0);
  if ((has_body && aqb > 119) || (has_head && aqh > 119)) atomicOr(p.err, VC2_DEVERR_QINDEX);
  const float fb = __uint_as_float(qt[min(aqb, 120)].w);
  const uint4 qh = qt[min(aqh, 120)]; // (index 120: a factor of 2^30 -- every 32-bit value quantises to zero)

  // ---- body: two strings of eight coefficients
  unsigned long long G0, G1;
  int L0, L1, last0, last1;
  float maxf = 0.f;
  p16_group(w0, fb, lut, G0, L0, last0, maxf);
  p16_group(w1, fb, lut, G1, L1, last1, maxf);
  int body_bits = L0 + L1, body_last = last1 ? L0 + last1 : last0;
  bool slow = maxf >= 32768.f /* an escape of the store */ || maxf * fb >= (float)(P16_MAXQ + 1) || max(L0, L1) > 63;
  if (!has_body) { body_bits = 0; body_last = 0; slow = false; }
  // ---- head: one coefficient, code by arithmetic
  unsigned hcode = 0;
  int hbits = 0;
  bool hnz = false;
  if (has_head) {
    // the deepest levels' coefficients are the ones that outgrow sixteen bits (and 2^20, where the float quotient stops
    // being exact: Fidelity's LL of a 12-bit picture after five levels reaches 1.5 million): an escape is fetched from the
    // wide array here and the quotient is the exact integer division of quant_core, one coefficient per lane; a code has
    // at most 32 bits (VLC.h:27) -- beyond that, the general coder raises the error
    if (hv == VC2_ST_SENTINEL) hv = *hwide;
    const int t = quant_core(hv, (int)qh.z, qh.x, (int)qh.y);
    slow |= (unsigned)(t + 65534) > 2u * 65534u;
    hcode = svlc_code(t);
    hbits = svlc_bits(t);
    hnz = t != 0;
  }

  const int cbr_total = CBR ? p.cbr_bytes[slice] : 0;
  bool bad_cbr = false;
  int bytes[3];
  auto comp_len = [&](int count) -> int { // ceil(bytes / scalar) by the rounded-up reciprocal (exact far beyond 255 * scalar)
    int len = (int)((float)(((count + 7) >> 3) + p.scalar - 1) * p.inv_scalar);
    if (len > 255) { atomicOr(p.err, VC2_DEVERR_SCALAR); len = 255; }
    return len * p.scalar;
  };
  auto cbr_v = [&](int need) -> int { // Slices.cpp:352-368: V absorbs the remainder of the slice
    if (!CBR) return need;
    const int vb = cbr_total - 4 - bytes[0] - bytes[1];
    if (vb < need) { if (lane == 0) atomicOr(p.err, VC2_DEVERR_CBR_TOOBIG); bad_cbr = true; return need; }
    if (vb / p.scalar > 255) { if (lane == 0) atomicOr(p.err, VC2_DEVERR_CBR_LEN); bad_cbr = true; return need; }
    return vb;
  };

  P16_STAT(0, true); P16_STAT(1, has_body && maxf >= 32768.f); P16_STAT(2, has_body && maxf * fb >= (float)(P16_MAXQ + 1));
  P16_STAT(3, has_body && max(L0, L1) > 63); P16_STAT(4, slow);
  // mode 2: a slice's byte count is known BEFORE its codes go into the image.  The four counts meet over a barrier; from then
  // on the workgroup's four slices share ONE image, back to back as they will lie in the payload (`wimg`, byte base `wb`:
  // the codes of neighbouring slices meet in a word through the LDS atomics that write them anyway), so that what leaves
  // the workgroup is one run of ~2.3 KB with one alignment.  One wavefront -- the LEADER, a different one from tile to
  // tile -- publishes the tile's count and requests the status words of the 64 tiles before this one; the answer travels
  // while all four write their codes
  unsigned long long lb_first = 0;
  bool lb_have = false;
  unsigned *wimg = img;
  int wb = 0;
  const int leader = LB ? (tile == (p.n_slices + NW - 1) / NW - 1 ? 0 : (tile & (NW - 1))) : 0; // (the last tile's later wavefronts may have no slice)
  auto lb_publish = [&](int total_bytes) {
    if (lane == 0) s_tot[wave] = total_bytes;
    __syncthreads();
    wimg = lds_u + P16_LUT_N + 4 * 128;
    { // (straight-line and through readfirstlane: as a loop over `wave` the base became a vector value and every address built
      // on it cost the kernel twenty registers -- two wavefronts per SIMD)
      int b = 0;
#pragma unroll
      for (int w2 = 0; w2 < NW - 1; ++w2) b += wave > w2 ? s_tot[w2] : 0;
      wb = __builtin_amdgcn_readfirstlane(b);
    }
    if (wave == leader && tile > 0) {
      unsigned long long *st = p.lookback + (size_t)pic * p.lookback_stride + 1;
      int agg = 0;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) agg += s_tot[w2];
      if (lane == 0) __hip_atomic_store(&st[tile], (1ull << 62) | (unsigned long long)agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int t = tile - 1 - lane;
      lb_first = t >= 0 ? __hip_atomic_load(&st[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
      lb_have = true;
    }
  };
  const bool slow_any = __any(slow);
  if (__builtin_expect(slow_any, 0)) {
    // the general coder, component by component: measure, then write (the image is still all zeros)
    const int32_t *recw = p.store_wide + rec_at;
    if constexpr (LB) { // (the slice's place in the tile's image follows from all three lengths: measure first)
#pragma unroll 1
      for (int cc = 0; cc < 3; ++cc)
        bytes[cc] = comp_len(p16_general<false>(rec + p.comp_off[cc], recw + p.comp_off[cc], p.comp_n[cc], p.comp_n0[cc], q, p.qmatrix, lane, nullptr, 0, 0, p.err));
      lb_publish(p.prefix + 4 + bytes[0] + bytes[1] + bytes[2]);
    }
    int base = p.prefix + 1;
    for (int cc = 0; cc < 3; ++cc) {
      const int16_t *src = rec + p.comp_off[cc];
      const int32_t *srcw = recw + p.comp_off[cc];
      if constexpr (!LB) {
        const int count = p16_general<false>(src, srcw, p.comp_n[cc], p.comp_n0[cc], q, p.qmatrix, lane, nullptr, 0, 0, p.err);
        bytes[cc] = comp_len(count);
        if (cc == 2) bytes[2] = cbr_v(bytes[2]);
      }
      p16_general<true>(src, srcw, p.comp_n[cc], p.comp_n0[cc], q, p.qmatrix, lane, wimg, 8 * (wb + base + 1), 8 * bytes[cc], p.err);
      if (lane == 0) put_byte(wimg, wb + base, (unsigned)(bytes[cc] / p.scalar));
      base += 1 + bytes[cc];
    }
    if (lane == 0) put_byte(wimg, wb + p.prefix, (unsigned)q & 0xFFu);
  } else {
    // ---- positions: one scan over (head bits << 16 | body bits), segments = components (rows 0-1, row 2, row 3)
    const int pk = (hbits << 16) | body_bits;
    int s = pk;
    s += dpp0<0x111, 0xf>(s);
    s += dpp0<0x112, 0xf>(s);
    s += dpp0<0x114, 0xf>(s);
    s += dpp0<0x118, 0xf>(s);
    s += dpp0<0x142, 0x2>(s); // row 0's total into row 1: luma spans two rows
    const int tot_y = __builtin_amdgcn_readlane(s, 31), tot_u = __builtin_amdgcn_readlane(s, 47), tot_v = __builtin_amdgcn_readlane(s, 63);
    const int tot = comp == 0 ? tot_y : (comp == 1 ? tot_u : tot_v);
    const int excl = s - pk;
    const int hpos = excl >> 16, bpos = (tot >> 16) + (excl & 0xFFFF); // bit offsets inside the component's data
    // bits through the last non-zero coefficient (component_slice_bytes' count): the body lies behind the head and
    // positions grow with the lane inside either, so the last lane with a non-zero body coefficient decides, or -- a
    // component whose body is all zeros -- the last lane with a non-zero head coefficient
    const int endpos = body_last ? bpos + body_last : (hnz ? hpos + hbits : 0);
    const unsigned long long nzb = __ballot(body_last != 0), nzh = __ballot(hnz);
    auto count_of = [&](unsigned b, unsigned h, int lane0) -> int {
      const unsigned m = b ? b : h;
      return m ? __builtin_amdgcn_readlane(endpos, lane0 + 31 - __builtin_clz(m)) : 0;
    };
    const int cnt_y = count_of((unsigned)nzb, (unsigned)nzh, 0);
    const int cnt_u = count_of((unsigned)(nzb >> 32) & 0xFFFFu, (unsigned)(nzh >> 32) & 0xFFFFu, 32);
    const int cnt_v = count_of((unsigned)(nzb >> 48), (unsigned)(nzh >> 48), 48);
    // the three lengths side by side on lanes 0-2, back into scalars
    const int len = comp_len(lane == 0 ? cnt_y : (lane == 1 ? cnt_u : cnt_v));
    bytes[0] = __builtin_amdgcn_readlane(len, 0);
    bytes[1] = __builtin_amdgcn_readlane(len, 1);
    bytes[2] = cbr_v(__builtin_amdgcn_readlane(len, 2));
    if constexpr (LB) lb_publish(p.prefix + 4 + bytes[0] + bytes[1] + bytes[2]);
    // data of component c starts one byte (its length byte) behind the previous component's end
    const int at_u = p.prefix + 2 + bytes[0], at_v = at_u + 1 + bytes[1]; // the length bytes of U and V
    const int bit0 = 8 * (wb + (comp == 0 ? p.prefix + 1 : (comp == 1 ? at_u : at_v)) + 1);
    const int room = 8 * (comp == 0 ? bytes[0] : (comp == 1 ? bytes[1] : bytes[2])); // bits of the component's data
    // a head code is inside the length or wholly beyond it (beyond the last non-zero coefficient every code is one bit)
    if (hbits && hpos + hbits <= room) {
      const unsigned v = hcode << (32 - hbits);
      unsigned *at = wimg + ((bit0 + hpos) >> 5);
      const unsigned bo = (unsigned)(bit0 + hpos);
      atomicOr(at, __builtin_amdgcn_alignbit(0u, v, bo));
      atomicOr(at + 1, __builtin_amdgcn_alignbit(v, 0u, bo));
    }
    if (has_body) {
      const int keep = min(max(room - bpos, 0), body_bits), k0 = min(keep, L0);
      p16_put(wimg, bit0 + bpos, G0, L0, k0);
      p16_put(wimg, bit0 + bpos + L0, G1, L1, keep - k0);
    }
    if (lane < 4) { // the quantiser index and the three length bytes
      const int at = lane == 0 ? p.prefix : (lane == 1 ? p.prefix + 1 : (lane == 2 ? at_u : at_v));
      const int lb = lane == 1 ? bytes[0] : (lane == 2 ? bytes[1] : bytes[2]);
      const unsigned val = lane == 0 ? ((unsigned)q & 0xFFu) : (unsigned)((float)lb * p.inv_scalar);
      put_byte(wimg, wb + at, val);
    }
  }
  const int total = p.prefix + 4 + bytes[0] + bytes[1] + bytes[2];
  wave_lds_sync();
  if constexpr (LB) {
    __syncthreads(); // the tile's image is complete
    // Only the LEADER waits for the look-back and moves the tile: the other three wavefronts are done and give their places to
    // the next workgroups' (with all four waiting the kernel took 1.88 ms per 128 UHD pictures against 1.45 + 0.51 for slots and
    // compaction: the wait -- a trip of the 64 status words to memory and back -- held four wavefront places per workgroup
    // for a third of their lives)
    if (wave != leader) return;
    unsigned long long *st = p.lookback + (size_t)pic * p.lookback_stride + 1;
    const unsigned long long M62 = (1ull << 62) - 1;
    unsigned long long run = 0;
    if (tile > 0) {
      unsigned long long v = lb_first;
      bool have = lb_have;
      int spins = 0;
      for (int base = tile - 1;;) {
        const int t = base - lane;
        // tiles before the first one count as an inclusive prefix of zero
        if (!have) v = t >= 0 ? __hip_atomic_load(&st[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
        have = false;
        const unsigned flag = (unsigned)(v >> 62);
        const unsigned long long m2 = __ballot(flag == 2), m0 = __ballot(flag == 0);
        const int first2 = m2 ? __ffsll((long long)m2) - 1 : 64;
        const unsigned long long need = first2 >= 63 ? ~0ull : ((2ull << first2) - 1);
        if ((m0 & need) == 0) { // every tile up to the first inclusive prefix has published
          run += (unsigned long long)wave_sum64((long long)(lane <= first2 ? (v & M62) : 0ull));
          if (first2 < 64) break;
          base -= 64;
          spins = 0;
          continue;
        }
        __builtin_amdgcn_s_sleep(1); // (0 / 1 / 8 and a fixed leader measured alike: 1.83 - 1.88 ms on the bench's pictures)
        if (++spins > (1 << 22)) { if (lane == 0) atomicOr(p.err, VC2_DEVERR_STREAM); break; } // (never: see the grid's order)
      }
    }
    int tot = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) tot += s_tot[w2];
    if (lane == 0) {
      const unsigned long long incl = run + (unsigned long long)tot;
      __hip_atomic_store(&st[tile], (2ull << 62) | incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tile == (p.n_slices + NW - 1) / NW - 1) p.lens[pic] = incl;
    }
    // bytes of the image in stream order: byte i = img[i >> 2] >> (24 - 8 * (i & 3)).  Byte stores for the ragged head and
    // tail of the destination, sixteen bytes per lane for its dword-aligned middle: destination dword w = stream bytes
    // head + 4 w ..., i. e. image words w and w + 1 shifted by `head` bytes (k_compact's arithmetic, from LDS)
    uint8_t *dst = p.payload + (size_t)pic * p.payload_stride + run;
    const unsigned *im = wimg;
    const int head = min((int)((4 - ((size_t)dst & 3)) & 3), tot);
    const int nw = (tot - head) >> 2, tail0 = head + 4 * nw;
    if (lane < head) dst[lane] = (uint8_t)(im[lane >> 2] >> (24 - 8 * (lane & 3)));
    if (lane < tot - tail0) { const int i = tail0 + lane; dst[i] = (uint8_t)(im[i >> 2] >> (24 - 8 * (i & 3))); }
    unsigned *d4 = (unsigned *)(dst + head);
    const unsigned h = (unsigned)head;
    for (int q4 = 4 * lane; q4 < nw; q4 += 256) {
      const uint4 v = *(const uint4 *)(im + q4);
      const unsigned nx = im[q4 + 4]; // (behind a tile of the greatest length: the images' guard words)
      const unsigned b0 = __builtin_bswap32(v.x), b1 = __builtin_bswap32(v.y), b2 = __builtin_bswap32(v.z), b3 = __builtin_bswap32(v.w), b4 = __builtin_bswap32(nx);
      Dword4 o;
      o.x = __builtin_amdgcn_alignbyte(b1, b0, h);
      o.y = __builtin_amdgcn_alignbyte(b2, b1, h);
      o.z = __builtin_amdgcn_alignbyte(b3, b2, h);
      o.w = __builtin_amdgcn_alignbyte(b4, b3, h);
      if (q4 + 4 <= nw) *(Dword4 *)(d4 + q4) = o;
      else {
        d4[q4] = o.x;
        if (q4 + 1 < nw) d4[q4 + 1] = o.y;
        if (q4 + 2 < nw) d4[q4 + 2] = o.z;
      }
    }
  } else if (CBR) {
    if (bad_cbr) return;
    uint8_t *dst = p.payload + (size_t)pic * p.payload_stride + p.cbr_offsets[slice];
    // dword stores for the 4-byte aligned middle of the destination (the budgets put a slice at any byte), byte stores for
    // its ragged head and tail
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
  } else {
    uint8_t *dst = p.slots + ((size_t)pic * p.n_slices + slice) * p.slot_bytes; // slots are whole 16-byte pieces
    if (lane == 0) p.sizes[(size_t)pic * p.n_slices + slice] = (unsigned)total;
    for (int i = lane * 16; i < total; i += 64 * 16) {
      const uint4 v = *(const uint4 *)((const uint8_t *)img + i);
      *(uint4 *)(dst + i) = make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// k_hq_pack16w: the same coder for LARGE slices -- a wavefront per COMPONENT, a workgroup (three wavefronts) per slice.
// A component of up to 64 head coefficients + 64 runs of sixteen (1024 + 64 coefficients: 32 x 32 slices at depth 5,
// UHD-2 4:4:4) does not fit the 32 / 16 / 16 lanes k_hq_pack16 gives it.  Here every wavefront codes its own component
// into the slice's one image; the three bit counts meet in LDS over a barrier (a component starts where the one before it
// ends), and each wavefront decides on its own whether its component takes the table path or the general coder.
//   lane16[64 c + lane] = matrix entry of the lane's body run | of its head coefficient << 8;  lane16[192 + c] = head[c],
//   lane16[195 + c] = body lanes of component c
static bool pack16w_plan(const PackParams &p, unsigned *lane16) {
  if (!p.store16 || !p.quantise || p.lookback || p.tile_slices) return false;
  if ((p.slice_coefs & 7) || (p.store_stride & 7)) return false; // every record on a 16-byte boundary (the uint4 loads)
  for (int l = 0; l < 200; ++l) lane16[l] = 0;
  int body_lanes = 0;
  for (int c = 0; c < 3; ++c) {
    const int n = p.comp_n[c], n0 = p.comp_n0[c];
    if (n <= 0 || n0 <= 0 || (p.comp_off[c] & 7)) return false;
    int start = 0, head = -1;
    for (int b = 0; b < 3 * p.depth + 1; ++b) {
      const int size = b == 0 ? n0 : n0 << (2 * ((b - 1) / 3));
      if (p.qmatrix[b] < 0 || p.qmatrix[b] > 255) return false;
      if (head < 0 && (size & 15) == 0 && (start & 15) == 0) head = start;
      if (head < 0) {
        if (start + size > 64) return false;
        for (int j = start; j < start + size; ++j) lane16[64 * c + j] |= (unsigned)p.qmatrix[b] << 8;
      } else {
        if ((start + size - head) / 16 > 64) return false;
        for (int j = start; j < start + size; j += 16) lane16[64 * c + (j - head) / 16] |= (unsigned)p.qmatrix[b];
      }
      start += size;
    }
    if (start != n) return false;
    if (head < 0) head = n;
    if (head & 7) return false;
    lane16[192 + c] = (unsigned)head;
    lane16[195 + c] = (unsigned)((n - head) / 16);
    body_lanes += (n - head) / 16;
  }
  return body_lanes >= 120; // (smaller slices: k_hq_pack16 or k_hq_pack)
}

template <bool CBR>
__global__ __launch_bounds__(192) void k_hq_pack16w(const PackParams p) {
  extern __shared__ unsigned lds_u[];
  __shared__ int s_cnt[3];
  const int lane = threadIdx.x & 63, comp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wavefront = component
  const int pic = blockIdx.y, slice = blockIdx.x;
  const int img_q = ((p.prefix + 4 + 3 * 255 * p.scalar + 3) / 4 + 2 + 3) / 4; // image size in 16-byte pieces
  unsigned *lut = lds_u;
  uint4 *qt = (uint4 *)(lds_u + P16_LUT_N);
  unsigned *img = lds_u + P16_LUT_N + 4 * 128;
  unsigned lut_e[2];
  lut_e[0] = g_vlc_lut_s[threadIdx.x];
  lut_e[1] = threadIdx.x < P16_LUT_N - 192 ? g_vlc_lut_s[192 + threadIdx.x] : 0u;
  const uint4 qt_e = g_p16_qt[threadIdx.x & 127];
  const unsigned lt = p.lane16[64 * comp + lane];
  const int head_n = (int)p.lane16[192 + comp], body_n = (int)p.lane16[195 + comp], coff = p.comp_off[comp];
  const bool has_body = lane < body_n, has_head = lane < head_n;
  const size_t rec_at = (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs;
  const int16_t *rec = (const int16_t *)p.store + rec_at;
  uint4 w0 = make_uint4(0u, 0u, 0u, 0u), w1 = w0;
  int hv = 0;
  if (has_body) { const int16_t *b = rec + coff + head_n + 16 * lane; w0 = *(const uint4 *)b; w1 = *(const uint4 *)(b + 8); }
  if (has_head) hv = rec[coff + lane];
  const int32_t *hwide = p.store_wide + rec_at + coff + lane;
  lut[threadIdx.x] = lut_e[0];
  if (threadIdx.x < P16_LUT_N - 192) lut[192 + threadIdx.x] = lut_e[1];
  if (threadIdx.x < 128) qt[threadIdx.x] = qt_e;
  const int q = p.qidx[(size_t)pic * p.n_slices + slice];
  for (int i = threadIdx.x; i < img_q; i += 192) ((uint4 *)img)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();

  const int aqb = max(q - (int)(lt & 0xFFu), 0), aqh = max(q - (int)((lt >> 8) & 0xFFu), 0);
  if ((has_body && aqb > 119) || (has_head && aqh > 119)) atomicOr(p.err, VC2_DEVERR_QINDEX);
  const float fb = __uint_as_float(qt[min(aqb, 120)].w);
  const uint4 qh = qt[min(aqh, 120)]; // (index 120: a factor of 2^30 -- every 32-bit value quantises to zero)
  unsigned long long G0, G1;
  int L0, L1, last0, last1;
  float maxf = 0.f;
  p16_group(w0, fb, lut, G0, L0, last0, maxf);
  p16_group(w1, fb, lut, G1, L1, last1, maxf);
  int body_bits = L0 + L1, body_last = last1 ? L0 + last1 : last0;
  bool slow = maxf >= 32768.f || maxf * fb >= (float)(P16_MAXQ + 1) || max(L0, L1) > 63;
  if (!has_body) { body_bits = 0; body_last = 0; slow = false; }
  unsigned hcode = 0;
  int hbits = 0;
  bool hnz = false;
  if (has_head) {
    // the deepest levels' coefficients are the ones that outgrow sixteen bits (and 2^20, where the float quotient stops
    // being exact: Fidelity's LL of a 12-bit picture after five levels reaches 1.5 million): an escape is fetched from the
    // wide array here and the quotient is the exact integer division of quant_core, one coefficient per lane; a code has
    // at most 32 bits (VLC.h:27) -- beyond that, the general coder raises the error
    if (hv == VC2_ST_SENTINEL) hv = *hwide;
    const int t = quant_core(hv, (int)qh.z, qh.x, (int)qh.y);
    slow |= (unsigned)(t + 65534) > 2u * 65534u;
    hcode = svlc_code(t);
    hbits = svlc_bits(t);
    hnz = t != 0;
  }
  P16_STAT(0, true); P16_STAT(1, has_body && maxf >= 32768.f); P16_STAT(2, has_body && maxf * fb >= (float)(P16_MAXQ + 1));
  P16_STAT(3, has_body && max(L0, L1) > 63); P16_STAT(4, slow);
  const bool general = __any(slow); // this component only
  const int16_t *src = rec + coff;
  const int32_t *srcw = p.store_wide + rec_at + coff;
  int hpos = 0, bpos = 0, count;
  if (__builtin_expect(general, 0)) {
    count = p16_general<false>(src, srcw, p.comp_n[comp], p.comp_n0[comp], q, p.qmatrix, lane, nullptr, 0, 0, p.err);
  } else {
    const int pk = (hbits << 16) | body_bits;
    const int s = seg_incl_scan<64>(pk, lane);
    const int tot = __builtin_amdgcn_readlane(s, 63);
    const int excl = s - pk;
    hpos = excl >> 16;
    bpos = (tot >> 16) + (excl & 0xFFFF);
    const int endpos = body_last ? bpos + body_last : (hnz ? hpos + hbits : 0);
    const unsigned long long nzb = __ballot(body_last != 0), nzh = __ballot(hnz);
    const unsigned long long m = nzb ? nzb : nzh;
    count = m ? __builtin_amdgcn_readlane(endpos, 63 - __builtin_clzll(m)) : 0;
  }
  if (lane == 0) s_cnt[comp] = count;
  __syncthreads();
  bool bad_cbr = false;
  int bytes[3];
  auto comp_len = [&](int cnt) -> int {
    int len = (int)((float)(((cnt + 7) >> 3) + p.scalar - 1) * p.inv_scalar);
    if (len > 255) { atomicOr(p.err, VC2_DEVERR_SCALAR); len = 255; }
    return len * p.scalar;
  };
  bytes[0] = comp_len(s_cnt[0]);
  bytes[1] = comp_len(s_cnt[1]);
  bytes[2] = comp_len(s_cnt[2]);
  if (CBR) { // Slices.cpp:352-368: V absorbs the remainder of the slice
    const int vb = p.cbr_bytes[slice] - 4 - bytes[0] - bytes[1];
    if (vb < bytes[2]) { if (threadIdx.x == 0) atomicOr(p.err, VC2_DEVERR_CBR_TOOBIG); bad_cbr = true; }
    else if (vb / p.scalar > 255) { if (threadIdx.x == 0) atomicOr(p.err, VC2_DEVERR_CBR_LEN); bad_cbr = true; }
    else bytes[2] = vb;
  }
  const int len_at = p.prefix + 1 + (comp > 0 ? 1 + bytes[0] : 0) + (comp > 1 ? 1 + bytes[1] : 0);
  const int mine = comp == 0 ? bytes[0] : (comp == 1 ? bytes[1] : bytes[2]);
  const int bit0 = 8 * (len_at + 1), room = 8 * mine;
  if (general) {
    p16_general<true>(src, srcw, p.comp_n[comp], p.comp_n0[comp], q, p.qmatrix, lane, img, bit0, room, p.err);
  } else {
    if (hbits && hpos + hbits <= room) {
      const unsigned v = hcode << (32 - hbits);
      unsigned *at = img + ((bit0 + hpos) >> 5);
      const unsigned bo = (unsigned)(bit0 + hpos);
      atomicOr(at, __builtin_amdgcn_alignbit(0u, v, bo));
      atomicOr(at + 1, __builtin_amdgcn_alignbit(v, 0u, bo));
    }
    if (has_body) {
      const int keep = min(max(room - bpos, 0), body_bits), k0 = min(keep, L0);
      p16_put(img, bit0 + bpos, G0, L0, k0);
      p16_put(img, bit0 + bpos + L0, G1, L1, keep - k0);
    }
  }
  if (lane == 0) put_byte(img, len_at, (unsigned)((float)mine * p.inv_scalar));
  if (threadIdx.x == 1) put_byte(img, p.prefix, (unsigned)q & 0xFFu);
  const int total = p.prefix + 4 + bytes[0] + bytes[1] + bytes[2];
  __syncthreads();
  if (CBR) {
    if (bad_cbr) return;
    uint8_t *dst = p.payload + (size_t)pic * p.payload_stride + p.cbr_offsets[slice];
    const int t = threadIdx.x;
    const int head = min((int)((4 - ((size_t)dst & 3)) & 3), total);
    const int nw = (total - head) >> 2, tail0 = head + 4 * nw;
    if (t < head) dst[t] = (uint8_t)(img[t >> 2] >> (24 - 8 * (t & 3)));
    if (t < total - tail0) { const int i = tail0 + t; dst[i] = (uint8_t)(img[i >> 2] >> (24 - 8 * (i & 3))); }
    unsigned *d4 = (unsigned *)(dst + head);
    for (int w = t; w < nw; w += 192) {
      const int i0 = head + 4 * w;
      const unsigned lo = __builtin_bswap32(img[i0 >> 2]), hi = __builtin_bswap32(img[(i0 >> 2) + 1]);
      d4[w] = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(i0 & 3));
    }
  } else {
    uint8_t *dst = p.slots + ((size_t)pic * p.n_slices + slice) * p.slot_bytes;
    if (threadIdx.x == 0) p.sizes[(size_t)pic * p.n_slices + slice] = (unsigned)total;
    for (int i = threadIdx.x * 16; i < total; i += 192 * 16) {
      const uint4 v = *(const uint4 *)((const uint8_t *)img + i);
      *(uint4 *)(dst + i) = make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w));
    }
  }
}

constexpr size_t P16_GUARD_BYTES = 128; // behind the last image: where p16_put's zero words of an all-zero tail may land
static size_t pack16w_lds(int prefix, int scalar) {
  const size_t img_q = (((size_t)prefix + 4 + 3 * 255 * (size_t)scalar + 3) / 4 + 2 + 3) / 4;
  return img_q * 16 + P16_LUT_N * 4 + 128 * 16 + P16_GUARD_BYTES;
}
static size_t pack16_lds(int prefix, int scalar, int waves = 4) {
  const size_t img_q = (((size_t)prefix + 4 + 3 * 255 * (size_t)scalar + 3) / 4 + 2 + 3) / 4;
  return (size_t)waves * img_q * 16 + P16_LUT_N * 4 + 128 * 16 + P16_GUARD_BYTES;
}
