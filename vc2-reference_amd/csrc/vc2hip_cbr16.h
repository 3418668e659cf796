// k_cbr_search16 -- quantIndicesCBR (EncodeStream.cpp:73-125, yss refinement Quantisation.cpp:627-642) for the common slice
// geometry on the 16-bit store, in the lane layout of k_hq_pack16 (vc2hip_pack16.h) with runs of EIGHT coefficients.
// (Included by vc2hip_slices.hip behind k_cbr_search_reg, whose search it keeps: only what a measurement costs changes.)
//
// k_cbr_search_reg gives a lane eight consecutive luma and eight consecutive chroma coefficients; the lanes that hold the
// first coefficients of a component see several subbands, so EVERY lane looks every coefficient's quantiser constants up
// (a clamp and an LDS read per coefficient and trial: 16 of them per measurement).  Here, as in the slice coder, a
// component is split into a HEAD -- its leading subbands whose blocks are not whole runs of eight (LL and the deepest
// level(s): 8 luma / 16 chroma coefficients for 32 x 16 slices at depth 4), one coefficient per lane -- and a BODY of
// eight-coefficient runs that each lie inside one subband: a lane holds one luma run (lanes 0 .. runsY-1), one chroma run
// (U on lanes 0 .. runsC-1, V behind it) and one head coefficient (luma on lanes 0.., U on 32.., V on 48..), and reads three
// constants per trial.  Bits through the last non-zero coefficient come from two packed scans, two ballots per component
// and a v_readlane (heads lie in front of bodies, positions grow with the lane inside either).
#pragma once

// lane8[lane] = matrix entry of the lane's luma run | of its chroma run << 8 | of its head coefficient << 16
// lane8[64] = luma head, [65] = chroma head (U and V alike), [66] = luma runs, [67] = chroma runs per component
static bool cbr16_plan(const CbrParams &p, unsigned *lane8) {
  if (!p.store16 || p.comp_n[1] != p.comp_n[2] || p.comp_n0[1] != p.comp_n0[2]) return false;
  if ((p.slice_coefs & 7) || (p.store_stride & 7)) return false; // every record on a 16-byte boundary (the uint4 loads)
  for (int l = 0; l < 72; ++l) lane8[l] = 0;
  int heads[3], runs[3];
  for (int c = 0; c < 3; ++c) {
    const int n = p.comp_n[c], n0 = p.comp_n0[c];
    if (n <= 0 || n0 <= 0 || (p.comp_off[c] & 7)) return false;
    int start = 0, head = -1;
    for (int b = 0; b < p.n_bands; ++b) {
      const int size = b == 0 ? n0 : n0 << (2 * ((b - 1) / 3));
      if (p.qmatrix[b] < 0 || p.qmatrix[b] > 255) return false;
      if (head < 0 && (size & 7) == 0 && (start & 7) == 0) head = start;
      if (head < 0) {
        const int lo = c == 0 ? 0 : (c == 1 ? 32 : 48), width = c == 0 ? 32 : 16;
        if (start + size > width) return false;
        for (int j = start; j < start + size; ++j) lane8[lo + j] |= (unsigned)p.qmatrix[b] << 16;
      } else {
        for (int j = start; j < start + size; j += 8) {
          const int run = (j - head) / 8;
          if (c == 0) { if (run >= 64) return false; lane8[run] |= (unsigned)p.qmatrix[b]; }
          else {
            const int per = (n - head) / 8, lane = (c == 1 ? 0 : per) + run;
            if (2 * per > 64) return false;
            lane8[lane] |= (unsigned)p.qmatrix[b] << 8;
          }
        }
      }
      start += size;
    }
    if (start != n) return false;
    if (head < 0) head = n;
    if (head & 7) return false; // the runs' 16-byte loads
    heads[c] = head; runs[c] = (n - head) / 8;
  }
  if (heads[1] != heads[2] || runs[1] != runs[2]) return false;
  lane8[64] = (unsigned)heads[0]; lane8[65] = (unsigned)heads[1]; lane8[66] = (unsigned)runs[0]; lane8[67] = (unsigned)runs[1];
  return runs[0] >= 40 && runs[1] >= 20; // (small slices: the lanes would idle; k_cbr_search_reg)
}

#ifndef VC2_CBR16_WPE
#define VC2_CBR16_WPE 1
#endif
__global__ __launch_bounds__(256, VC2_CBR16_WPE) void k_cbr_search16(const CbrParams p) {
  __shared__ uint4 s_tab[80]; // by quantiser index: (rounded-up 4 / factor as a float, factor, offset + 2, -)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slice0 = (blockIdx.x * 4 + wave) * CBR_SPW, pic = blockIdx.y;
  if (threadIdx.x < 80)
    s_tab[threadIdx.x] = make_uint4(__float_as_uint(c_qs.inv4[threadIdx.x]), (unsigned)c_qs.qf[threadIdx.x], (unsigned)c_qs.off[threadIdx.x] + 2u, 0u);
  const unsigned lq = p.lane8[lane];
  const int headY = (int)p.lane8[64], headC = (int)p.lane8[65], runsY = (int)p.lane8[66], runsC = (int)p.lane8[67];
  __syncthreads();
  const bool has_y = lane < runsY, has_c = lane < 2 * runsC;
  const int ccb = lane < runsC ? 1 : 2, crun = lane < runsC ? lane : lane - runsC; // the chroma run's component, its number
  const int hc = lane < 32 ? 0 : (lane < 48 ? 1 : 2), hj = lane - (lane < 32 ? 0 : (lane < 48 ? 32 : 48)); // the head coefficient's
  const bool has_h = hj < (hc == 0 ? headY : headC);
  const int m_y = 16 * (int)(lq & 0xFFu), m_c = 16 * (int)((lq >> 8) & 0xFFu), m_h = 16 * (int)((lq >> 16) & 0xFFu);
  int guess = -1; // the previous slice's threshold
  for (int slice = slice0; slice < min(slice0 + CBR_SPW, p.n_slices); ++slice) { // no workgroup barriers below
  const size_t rec_at = (size_t)pic * p.store_stride + (size_t)slice * p.slice_coefs;
  const int16_t *rec = (const int16_t *)p.store + rec_at;
  float fy[8], fc[8], fh = 0.f; // |coefficient|
  bool out = false;
  {
    uint4 wy = make_uint4(0u, 0u, 0u, 0u), wc = wy;
    int hv = 0;
    if (has_y) wy = *(const uint4 *)(rec + p.comp_off[0] + headY + 8 * lane);
    if (has_c) wc = *(const uint4 *)(rec + p.comp_off[ccb] + headC + 8 * crun);
    if (has_h) hv = rec[p.comp_off[hc] + hj];
    const unsigned dy[4] = {wy.x, wy.y, wy.z, wy.w}, dc[4] = {wc.x, wc.y, wc.z, wc.w};
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      fy[2 * k] = __builtin_fabsf((float)(int)(short)(dy[k] & 0xFFFFu)); fy[2 * k + 1] = __builtin_fabsf((float)((int)dy[k] >> 16));
      fc[2 * k] = __builtin_fabsf((float)(int)(short)(dc[k] & 0xFFFFu)); fc[2 * k + 1] = __builtin_fabsf((float)((int)dc[k] >> 16));
      mx = fmaxf(mx, fmaxf(fmaxf(fy[2 * k], fy[2 * k + 1]), fmaxf(fc[2 * k], fc[2 * k + 1])));
    }
    fh = __builtin_fabsf((float)hv);
    out = fmaxf(mx, fh) > 32767.f; // an escape of the 16-bit store (the sentinel is -32768): the general kernel
  }
  const int avail = p.slice_bytes[slice] - 4;
  const char *tab = (const char *)s_tab;
  auto entry = [&](int tq16, int m) -> int { return min(max(tq16 - m, 0), 16 * 79); };
  // code lengths of eight magnitudes under one reciprocal r: total and the end of the last non-zero code (see k_cbr_search_reg)
  auto bits8u = [&](const float (&f)[8], float r, bool has, int &sum, int &last_end) {
    sum = 0; last_end = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int eb = (int)__builtin_amdgcn_ubfe(__float_as_uint(__builtin_fmaf(f[k], r, 1.0f)), 23, 8); // 127 + E
      sum += 2 * eb + min(eb, 128) - 380;                                                               // 2E + min(E, 1) + 1
      last_end = eb >= 128 ? sum : last_end;
    }
    if (!has) { sum = 0; last_end = 0; }
  };
  auto comp_bytes = [&](int count, bool &bad) -> int {
    const int len = (int)((float)(((count + 7) >> 3) + p.scalar - 1) * p.inv_scalar);
    bad |= len > 255;
    return __mul24(len, p.scalar);
  };
  auto last_of = [&](unsigned long long body, unsigned long long head, int v_body, int v_head) -> int { // the last lane with a non-zero coefficient decides
    if (body) return __builtin_amdgcn_readlane(v_body, 63 - __builtin_clzll(body));
    if (head) return __builtin_amdgcn_readlane(v_head, 63 - __builtin_clzll(head));
    return 0;
  };
  auto need_bytes = [&](int tq, bool &bad) -> int {
    bad |= tq - p.qm_min > 79; // a factor of 2^22 or more: outside the float domain
    const float ry = *(const float *)(tab + entry(16 * tq, m_y)), rc = *(const float *)(tab + entry(16 * tq, m_c)),
                rh = *(const float *)(tab + entry(16 * tq, m_h));
    int sy, ly, sc, lc;
    bits8u(fy, ry, has_y, sy, ly);
    bits8u(fc, rc, has_c, sc, lc);
    const int ebh = (int)__builtin_amdgcn_ubfe(__float_as_uint(__builtin_fmaf(fh, rh, 1.0f)), 23, 8);
    const int hb = has_h ? 2 * ebh + min(ebh, 128) - 380 : 0;
    const bool hnz = has_h && ebh >= 128;
    // luma: head bits (lanes 0 ..) << 16 | body bits; chroma: heads on lanes 32 .. 47 (U), 48 .. 63 (V), bodies U then V from lane 0
    const int pk1 = ((hc == 0 ? hb : 0) << 16) | sy, pk2 = ((hc != 0 ? hb : 0) << 16) | sc;
    const int s1 = wave_incl_scan(pk1, lane), s2 = wave_incl_scan(pk2, lane);
    const int e1 = s1 - pk1, e2 = s2 - pk2;
    const int tot_hy = __builtin_amdgcn_readlane(s1, 63) >> 16;
    const int tot_hu = __builtin_amdgcn_readlane(s2, 47) >> 16, tot_hv = (__builtin_amdgcn_readlane(s2, 63) >> 16) - tot_hu;
    const int tot_bu = __builtin_amdgcn_readlane(s2, runsC - 1) & 0xFFFF;
    // end of the lane's last non-zero code inside its component's data: body (behind the component's head) and head
    const int end_yb = tot_hy + (e1 & 0xFFFF) + ly, end_yh = (e1 >> 16) + hb;
    const int end_cb = (ccb == 1 ? tot_hu + (e2 & 0xFFFF) : tot_hv + (e2 & 0xFFFF) - tot_bu) + lc;
    const int end_ch = (e2 >> 16) - (hc == 2 ? tot_hu : 0) + hb;
    const unsigned long long b_y = __ballot(ly != 0), h_y = __ballot(hnz && hc == 0);
    const unsigned long long b_c = __ballot(lc != 0), h_c = __ballot(hnz && hc != 0);
    const unsigned long long u_lanes = runsC >= 64 ? ~0ull : ((1ull << runsC) - 1);
    int need = comp_bytes(last_of(b_y, h_y, end_yb, end_yh), bad);
    need += comp_bytes(last_of(b_c & u_lanes, h_c & 0x0000FFFF00000000ull, end_cb, end_ch), bad);
    need += comp_bytes(last_of(b_c & ~u_lanes, h_c & 0xFFFF000000000000ull, end_cb, end_ch), bad);
    return need;
  };
  // luma-only sum of squared reconstruction error (EncodeStream.cpp:73-125 through quant / scale, Quantisation.cpp:69-95)
  auto yss = [&](int tq, bool &bad) -> long long {
    bad |= tq - p.qm_min > 79;
    long long acc = 1ll << 35; // keeps the lane's sum of 32-bit products non-negative
    const uint4 ty = *(const uint4 *)(tab + entry(16 * tq, m_y)), th = *(const uint4 *)(tab + entry(16 * tq, m_h));
    auto err2 = [&](float f, const uint4 &t) -> long long {
      const unsigned q = (unsigned)(f * __uint_as_float(t.x));
      const unsigned r = (__umul24(q, t.y) + __umul24(min(q, 1u), t.z)) >> 2; // scale(): nothing is added to a zero
      const int d = (int)f - (int)r;
      return (long long)__mul24(d, d);
    };
    if (has_y) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += err2(fy[k], ty);
    }
    if (has_h && hc == 0) acc += err2(fh, th);
    const int lo = seg_incl_scan<64>((int)(acc & 0xFFFFFF), lane), hi = seg_incl_scan<64>((int)(acc >> 24), lane);
    return (long long)__builtin_amdgcn_readlane(lo, 63) + ((long long)__builtin_amdgcn_readlane(hi, 63) << 24) - (64ll << 35);
  };

  // ---- the search of k_cbr_search_reg, unchanged: threshold by monotonicity from the predecessor's, the reference's
  // smallest trial for the error it would raise, then the refinement by the luma error
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
      if (hi == 127) { if (lo >= 126) break; t = min(126, lo + step); step *= 2; }
      else if (lo < 0) { if (hi <= 0) break; t = max(0, hi - step); step *= 2; }
      else t = (lo + hi) >> 1;
    }
    q = hi;
    if (!bad) { // the smallest trial of the reference's walk to this threshold
      int rt = 63, rd = 64, rmin = 127, rmax = 0;
      while (rd > 0) { rd >>= 1; rmin = min(rmin, rt); rmax = max(rmax, rt); if (rt >= q) rt -= rd; else rt += rd; }
      if (rmax - p.qm_min > 79) bad = true;
      else if (rmin < lowest) { (void)need_bytes(rmin, bad); bad = __any(bad); }
    }
  }
  const int q_fit = q;
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
