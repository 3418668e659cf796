/*
 * vc2_oracle.c -- TEST INFRASTRUCTURE ONLY (see vc2_oracle.h).
 *
 * Scalar C restatement of the bbc/vc2-reference hot path.  Written from the
 * behaviour of the reference (file:line citations on every function), not from
 * its text: planes are flat row-major int32 arrays, subbands are addressed by
 * (stride, phase) in place, bit I/O works on memory buffers.
 */
#include "vc2_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];
const char *vc2o_last_error(void) { return g_err; }
static int fail(int code, const char *msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

/* ------------------------------------------------------------------------ */
/* geometry                                                                  */
/* ------------------------------------------------------------------------ */

/* WaveletTransform.cpp:74-77 */
int vc2o_padded_size(int size, int depth) {
  const int cell = 1 << depth;
  return cell * ((size + cell - 1) / cell);
}

/* WaveletTransform.cpp:116-136 (returns number of slices, 0 if invalid) */
int vc2o_slice_size_is_valid(int depth, int len_luma, int len_chroma, int n_size) {
  if (depth <= 0 || depth > 31) return 0;
  const int unit = 1 << depth;
  const int max_slices = (len_luma < len_chroma ? len_luma : len_chroma) / unit;
  if (n_size <= 0 || n_size > max_slices) return 0;
  const int transform_size = n_size * unit;
  const int pl = vc2o_padded_size(len_luma, depth);
  const int pc = vc2o_padded_size(len_chroma, depth);
  const int n = (pl + transform_size - 1) / transform_size;
  if (pl % n == 0 && (pl / n) % unit == 0 && pc % n == 0 && (pc / n) % unit == 0) return n;
  return 0;
}

/* Picture.cpp:49-73 */
void vc2o_chroma_dims(int h, int w, int cf, int *ch, int *cw) {
  *ch = (cf == VC2O_CF420) ? h / 2 : h;
  *cw = (cf == VC2O_CF444) ? w : w / 2;
}

/* ------------------------------------------------------------------------ */
/* sample I/O                                                                */
/* ------------------------------------------------------------------------ */

/* Arrays.cpp:333-379 with the EncodeStream.cpp:319-322 manipulators:
 * big-endian word, left justified (logical >> by 8*bytes-depth), offset binary. */
void vc2o_ingest(const uint8_t *raw, int word_bytes, int bit_depth, size_t n, int32_t *out) {
  const int shift = 8 * word_bytes - bit_depth;
  const int32_t offset = 1 << (bit_depth - 1);
  for (size_t i = 0; i < n; ++i) {
    uint32_t v = 0;
    for (int b = 0; b < word_bytes; ++b) v = (v << 8) | raw[i * word_bytes + b];
    v >>= shift;
    out[i] = (int32_t)v - offset;
  }
}

/* Picture.cpp:284-292 clip + Arrays.cpp:381-426 write (DecodeStream.cpp:591-605) */
void vc2o_clip_emit(const int32_t *in, size_t n, int word_bytes, int bit_depth, uint8_t *out) {
  const int shift = 8 * word_bytes - bit_depth;
  const int32_t lo = -(1 << (bit_depth - 1)), hi = (1 << (bit_depth - 1)) - 1;
  for (size_t i = 0; i < n; ++i) {
    int32_t s = in[i] < lo ? lo : (in[i] > hi ? hi : in[i]);
    uint32_t v = (uint32_t)(s - lo) << shift;
    for (int b = 0; b < word_bytes; ++b)
      out[i * word_bytes + b] = (uint8_t)(v >> (8 * (word_bytes - 1 - b)));
  }
}

/* ------------------------------------------------------------------------ */
/* wavelet transform                                                         */
/* ------------------------------------------------------------------------ */

/* WaveletTransform.cpp:79-94: replicate the last row / column */
void vc2o_pad(const int32_t *in, int h, int w, int32_t *out, int ph, int pw) {
  for (int y = 0; y < ph; ++y) {
    const int sy = y < h ? y : h - 1;
    for (int x = 0; x < pw; ++x) out[(size_t)y * pw + x] = in[(size_t)sy * w + (x < w ? x : w - 1)];
  }
}

/* One lifting step: x[t] += sign * ((sum_k w[k]*x[clamp(t+off[k])] + round) >> shift)
 * for every t of parity `odd`.  Tap indices clamp to the same-parity range:
 * even taps to [0,n-2], odd taps to [1,n-1] (WaveletTransform.cpp:498-533 etc.) */
typedef struct {
  int odd, ntaps, off[8], w[8], round, shift, sign;
} lift_step;
typedef struct {
  int nsteps, accuracy;
  lift_step s[4];
} wavelet_def;

#define PRED_DD                                                     \
  { 1, 4, {-3, -1, 1, 3}, {-1, 9, 9, -1}, 8, 4, -1 }
#define UPD_2TAP                                                    \
  { 0, 2, {-1, 1}, {1, 1}, 2, 2, +1 }
static const wavelet_def WAVELETS[7] = {
    /* DD97  :478-533 */ {2, 1, {PRED_DD, UPD_2TAP}},
    /* LeGall:595-644 */ {2, 1, {{1, 2, {-1, 1}, {1, 1}, 1, 1, -1}, UPD_2TAP}},
    /* DD137 :700-761 */ {2, 1, {PRED_DD, {0, 4, {-3, -1, 1, 3}, {-1, 9, 9, -1}, 16, 5, +1}}},
    /* Haar0 :829-871 */ {2, 0, {{1, 1, {-1}, {1}, 0, 0, -1}, {0, 1, {1}, {1}, 1, 1, +1}}},
    /* Haar1           */ {2, 1, {{1, 1, {-1}, {1}, 0, 0, -1}, {0, 1, {1}, {1}, 1, 1, +1}}},
    /* Fidelity :919-1001 (update first, then predict) */
    {2, 0,
     {{0, 8, {-7, -5, -3, -1, 1, 3, 5, 7}, {-8, 21, -46, 161, 161, -46, 21, -8}, 128, 8, +1},
      {1, 8, {-7, -5, -3, -1, 1, 3, 5, 7}, {-2, 10, -25, 81, 81, -25, 10, -2}, 128, 8, -1}}},
    /* Daub97 :1090-1175 */
    {4, 1,
     {{1, 2, {-1, 1}, {6497, 6497}, 2048, 12, -1},
      {0, 2, {-1, 1}, {217, 217}, 2048, 12, -1},
      {1, 2, {-1, 1}, {3616, 3616}, 2048, 12, +1},
      {0, 2, {-1, 1}, {1817, 1817}, 2048, 12, +1}}}};

static inline int clamp_tap(int i, int n) {
  if (i & 1) return i < 1 ? 1 : (i > n - 1 ? n - 1 : i);
  return i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
}

static void lift_line(int32_t *x, ptrdiff_t stride, int n, const lift_step *st, int invert) {
  const int sign = invert ? -st->sign : st->sign;
  for (int t = st->odd; t < n; t += 2) {
    int32_t sum = st->round;
    for (int k = 0; k < st->ntaps; ++k) sum += st->w[k] * x[clamp_tap(t + st->off[k], n) * stride];
    x[t * stride] += sign * (sum >> st->shift);
  }
}

/* WaveletTransform.cpp:262-281 + the per-kernel level functions */
int vc2o_dwt_forward(int32_t *p, int ph, int pw, int kernel, int depth) {
  if (kernel < 0 || kernel > 6) return fail(VC2O_EINVAL, "invalid wavelet kernel");
  const wavelet_def *wd = &WAVELETS[kernel];
  for (int level = 0; level < depth; ++level) {
    const int s = 1 << level, rows = ph / s, cols = pw / s;
    if (wd->accuracy)
      for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) p[(size_t)y * s * pw + (size_t)x * s] <<= wd->accuracy;
    for (int k = 0; k < wd->nsteps; ++k)
      for (int y = 0; y < rows; ++y) lift_line(p + (size_t)y * s * pw, s, cols, &wd->s[k], 0);
    for (int k = 0; k < wd->nsteps; ++k)
      for (int x = 0; x < cols; ++x) lift_line(p + (size_t)x * s, (ptrdiff_t)s * pw, rows, &wd->s[k], 0);
  }
  return 0;
}

/* WaveletTransform.cpp:321-342 (without the final crop) */
int vc2o_dwt_inverse(int32_t *p, int ph, int pw, int kernel, int depth) {
  if (kernel < 0 || kernel > 6) return fail(VC2O_EINVAL, "invalid wavelet kernel");
  const wavelet_def *wd = &WAVELETS[kernel];
  for (int level = depth - 1; level >= 0; --level) {
    const int s = 1 << level, rows = ph / s, cols = pw / s;
    for (int k = wd->nsteps - 1; k >= 0; --k)
      for (int x = 0; x < cols; ++x) lift_line(p + (size_t)x * s, (ptrdiff_t)s * pw, rows, &wd->s[k], 1);
    for (int k = wd->nsteps - 1; k >= 0; --k)
      for (int y = 0; y < rows; ++y) lift_line(p + (size_t)y * s * pw, s, cols, &wd->s[k], 1);
    if (wd->accuracy) {
      const int32_t r = 1 << (wd->accuracy - 1);
      for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
          int32_t *q = &p[(size_t)y * s * pw + (size_t)x * s];
          *q = (*q + r) >> wd->accuracy;
        }
    }
  }
  return 0;
}

/* WaveletTransform.cpp:345-423.  The reference evaluates this with float
 * variables, double pow() and float logf()/floorf() (math.h overloads). */
int vc2o_quant_matrix(int kernel, int depth, int32_t *out) {
  static const float ALPHA[7] = {1.280868846f, 1.224744871f, 1.280868846f, 1.414213562f,
                                 1.414213562f, 0.682408629f, 1.139917028f};
  static const float BETA[7] = {0.820572875f, 0.847791248f, 0.809253958f, 0.707106871f,
                                0.707106871f, 1.367856979f, 0.887168005f};
  static const int SHIFT[7] = {1, 1, 1, 0, 1, 0, 1};
  if (depth < 0) return fail(VC2O_EINVAL, "wavelet depth may not be < 0");
  if (kernel < 0 || kernel > 6) return fail(VC2O_EINVAL, "invalid wavelet kernel");
  if (depth == 0) { out[0] = 0; return 0; }
  const float alpha = ALPHA[kernel], beta = BETA[kernel];
  const int shift = SHIFT[kernel];
  const float a2 = alpha * alpha, ab = alpha * beta, b2 = beta * beta;
  float ll[32], lh[32], hh[32], min_gain = FLT_MAX;
  for (int level = depth; level > 0; --level) {
    const float scale = (float)(pow((double)a2, depth - level) / pow(2.0, shift * (depth - level + 1)));
    ll[level] = scale * a2;
    lh[level] = scale * ab;
    hh[level] = scale * b2;
    float m = ll[level] < lh[level] ? ll[level] : lh[level];
    m = m < hh[level] ? m : hh[level];
    min_gain = m < min_gain ? m : min_gain;
  }
  int idx = 0;
  out[idx++] = (int)floorf(4.0f * logf(ll[1] / min_gain) / logf(2.0f) + 0.5f);
  for (int level = 1; level <= depth; ++level) {
    const int l = (int)floorf(4.0f * logf(lh[level] / min_gain) / logf(2.0f) + 0.5f);
    const int h = (int)floorf(4.0f * logf(hh[level] / min_gain) / logf(2.0f) + 0.5f);
    out[idx++] = l;
    out[idx++] = l;
    out[idx++] = h;
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* quantiser                                                                 */
/* ------------------------------------------------------------------------ */

/* Quantisation.cpp:40-66.  The reference holds SMPTE 2042-1's quant_factor()
 * as a 120-entry table; the entries are generated here from the standard's
 * closed form (tests/test_oracle.py checks all 120 against the reference file
 * when /root/reference is present).  Entries 116..119 exceed INT_MAX and are
 * narrowed to int exactly like the reference's static_cast<int>. */
int vc2o_quant_factor(int q, int32_t *out) {
  if (q > 119) return fail(VC2O_EQINDEX, "quantization index exceeds maximum implemented value.");
  if (q < 0) q = 0;
  const uint64_t base = 1ull << (q / 4);
  uint64_t f;
  switch (q % 4) {
    case 0: f = 4 * base; break;
    case 1: f = (503829 * base + 52958) / 105917; break;
    case 2: f = (665857 * base + 58854) / 117708; break;
    default: f = (440253 * base + 32722) / 65444; break;
  }
  *out = (int32_t)(uint32_t)f;
  return 0;
}

/* Quantisation.cpp:78-83 */
static int quant_offset(int q, int32_t *out) {
  if (q < 0) q = 0;
  if (q == 0) { *out = 1; return 0; }
  if (q == 1) { *out = 2; return 0; }
  int32_t qf;
  const int rc = vc2o_quant_factor(q, &qf);
  if (rc) return rc;
  *out = (int32_t)(((uint32_t)qf + 1u)) / 2; /* (qf+1)/2 in int, wraps like the reference */
  return 0;
}

/* Quantisation.cpp:69-76 */
int vc2o_quant(int32_t v, int aq, int32_t *out) {
  int32_t qf;
  const int rc = vc2o_quant_factor(aq, &qf);
  if (rc) return rc;
  const int neg = v < 0;
  int32_t a = neg ? (int32_t)(0u - (uint32_t)v) : v;
  a = (int32_t)((uint32_t)a << 2);
  a /= qf;
  *out = neg ? (int32_t)(0u - (uint32_t)a) : a;
  return 0;
}

/* Quantisation.cpp:86-95 */
int vc2o_scale(int32_t v, int aq, int32_t *out) {
  int32_t qf, off;
  int rc = vc2o_quant_factor(aq, &qf);
  if (rc) return rc;
  rc = quant_offset(aq, &off);
  if (rc) return rc;
  const int neg = v < 0;
  int32_t a = neg ? (int32_t)(0u - (uint32_t)v) : v;
  a = (int32_t)((uint32_t)a * (uint32_t)qf);
  if (a > 0) a = (int32_t)((uint32_t)a + (uint32_t)off);
  a = (int32_t)((uint32_t)a + 2u);
  a /= 4;
  *out = neg ? (int32_t)(0u - (uint32_t)a) : a;
  return 0;
}

/* Subband b of an in-place transform: (stride, y phase, x phase).
 * WaveletTransform.cpp:428-450; order LL, then per level HL, LH, HH. */
static void band_geom(int band, int depth, int *stride, int *oy, int *ox) {
  if (band == 0) { *stride = 1 << depth; *oy = 0; *ox = 0; return; }
  const int level = (band - 1) / 3 + 1, kind = (band - 1) % 3;
  *stride = 1 << (depth + 1 - level);
  const int o = *stride / 2;
  *oy = (kind == 0) ? 0 : o; /* HL: row phase 0 */
  *ox = (kind == 1) ? 0 : o; /* LH: col phase 0 */
}

/* band index of plane position (y,x) */
static int band_of(int y, int x, int depth) {
  for (int level = depth; level >= 1; --level) { /* finest first: stride 2 */
    const int s = 1 << (depth + 1 - level), o = s / 2;
    const int ym = y % s, xm = x % s;
    if (ym == 0 && xm == o) return 3 * (level - 1) + 1;
    if (ym == o && xm == 0) return 3 * (level - 1) + 2;
    if (ym == o && xm == o) return 3 * (level - 1) + 3;
  }
  return 0;
}

static inline int adjust_q(int q, int m) { return q - m < 0 ? 0 : q - m; } /* Quantisation.cpp:16-20 */

typedef int (*qfun)(int32_t, int, int32_t *);

/* Quantisation.cpp:386-428 / :433-475: each subband is cut into ys x xs blocks at
 * (i+1)*dim/n and each block uses its slice's adjusted index. */
static int apply_np(qfun f, const int32_t *in, int ph, int pw, int depth, const int32_t *qidx,
                    int ys, int xs, const int32_t *qm, int32_t *out, int skip_ll) {
  for (int band = skip_ll ? 1 : 0; band < 3 * depth + 1; ++band) {
    int s, oy, ox;
    band_geom(band, depth, &s, &oy, &ox);
    const int bh = ph / s, bw = pw / s;
    for (int by = 0; by < ys; ++by)
      for (int bx = 0; bx < xs; ++bx) {
        const int aq = adjust_q(qidx[by * xs + bx], qm[band]);
        const int top = by * bh / ys, bottom = (by + 1) * bh / ys;
        const int left = bx * bw / xs, right = (bx + 1) * bw / xs;
        for (int y = top; y < bottom; ++y)
          for (int x = left; x < right; ++x) {
            const size_t i = (size_t)(y * s + oy) * pw + (size_t)(x * s + ox);
            const int rc = f(in[i], aq, &out[i]);
            if (rc) return rc;
          }
      }
  }
  return 0;
}

int vc2o_quantise_np(const int32_t *coef, int ph, int pw, int depth, const int32_t *qidx, int ys,
                     int xs, const int32_t *qm, int32_t *out) {
  return apply_np(vc2o_quant, coef, ph, pw, depth, qidx, ys, xs, qm, out, 0);
}
int vc2o_dequantise_np(const int32_t *q, int ph, int pw, int depth, const int32_t *qidx, int ys,
                       int xs, const int32_t *qm, int32_t *out) {
  return apply_np(vc2o_scale, q, ph, pw, depth, qidx, ys, xs, qm, out, 0);
}

/* Quantisation.cpp:191-208 */
static int32_t predict_dc(const int32_t *ll, int w, int y, int x) {
  if (y > 0 && x > 0) {
    const int32_t r = ll[(y - 1) * w + x - 1] + ll[(y - 1) * w + x] + ll[y * w + x - 1];
    return r >= 0 ? (r + 1) / 3 : (r - 1) / 3;
  }
  if (y > 0) return ll[(y - 1) * w + x];
  if (x > 0) return ll[y * w + x - 1];
  return 0;
}

/* Quantisation.cpp:213-234 (encode) / :287-306 (decode): raster scan of the whole LL band */
static int ll_predicted(int encode, const int32_t *in, int ph, int pw, int depth,
                        const int32_t *qidx, int ys, int xs, int qm0, int32_t *out) {
  const int s = 1 << depth, lh = ph / s, lw = pw / s;
  int32_t *restored = (int32_t *)malloc(sizeof(int32_t) * (size_t)lh * lw);
  int rc = 0;
  for (int y = 0; y < lh && !rc; ++y)
    for (int x = 0; x < lw; ++x) {
      const int yb = ((y + 1) * ys - 1) / lh, xb = ((x + 1) * xs - 1) / lw;
      const int aq = adjust_q(qidx[yb * xs + xb], qm0);
      const int32_t pred = predict_dc(restored, lw, y, x);
      const size_t i = (size_t)y * s * pw + (size_t)x * s;
      int32_t q, r;
      if (encode) {
        if ((rc = vc2o_quant(in[i] - pred, aq, &q))) break;
        out[i] = q;
      } else {
        q = in[i];
      }
      if ((rc = vc2o_scale(q, aq, &r))) break;
      restored[y * lw + x] = r + pred;
      if (!encode) out[i] = r + pred;
    }
  free(restored);
  return rc;
}

int vc2o_quantise_ld(const int32_t *coef, int ph, int pw, int depth, const int32_t *qidx, int ys,
                     int xs, const int32_t *qm, int32_t *out) {
  int rc = ll_predicted(1, coef, ph, pw, depth, qidx, ys, xs, qm[0], out);
  if (rc) return rc;
  return apply_np(vc2o_quant, coef, ph, pw, depth, qidx, ys, xs, qm, out, 1);
}
int vc2o_dequantise_ld(const int32_t *q, int ph, int pw, int depth, const int32_t *qidx, int ys,
                       int xs, const int32_t *qm, int32_t *out) {
  int rc = ll_predicted(0, q, ph, pw, depth, qidx, ys, xs, qm[0], out);
  if (rc) return rc;
  return apply_np(vc2o_scale, q, ph, pw, depth, qidx, ys, xs, qm, out, 1);
}

/* ------------------------------------------------------------------------ */
/* bit I/O (VLC.cpp:98-257) on memory buffers                                */
/* ------------------------------------------------------------------------ */

typedef struct {
  uint8_t *buf;
  size_t cap, pos;
  unsigned cache;
  int cached, bounded, err;
  long left;
} bitw;

/* VLC.cpp:151-172 */
static void put_bit(bitw *w, int bit) {
  if (w->bounded && w->left < 1) {
    if (bit) return;
    w->err = VC2O_EBOUNDED;
    return;
  }
  w->cache = ((w->cache << 1) | (bit ? 1u : 0u)) & 0xFFu;
  ++w->cached;
  --w->left;
  if (w->cached == 8) {
    if (w->pos < w->cap) w->buf[w->pos] = (uint8_t)w->cache; else w->err = VC2O_ECAP;
    ++w->pos;
    w->cached = 0;
  }
}
static void put_bits(bitw *w, unsigned n, uint32_t v) { while (n > 0) { --n; put_bit(w, (v >> n) & 1u); } }
static void w_bounded(bitw *w, long bits) { w->bounded = 1; w->left = bits; }
static void w_flush(bitw *w) { if (w->bounded) while (w->left > 0) put_bit(w, 0); } /* :229-235 */
static void w_align(bitw *w) { w->bounded = 0; while (w->cached) put_bit(w, 0); }  /* :246-250 */
static void put_bytes(bitw *w, int n, uint32_t v) { /* :326-335 */
  w_align(w);
  while (n > 0) {
    --n;
    if (w->pos < w->cap) w->buf[w->pos] = (uint8_t)(v >> (8 * n)); else w->err = VC2O_ECAP;
    ++w->pos;
  }
}

/* VLC.cpp:21-52: interleaved exp-Golomb; returns code and length */
static unsigned uvlc_code(uint32_t value, uint32_t *bits) {
  if (value == 0) { *bits = 1; return 1; }
  value += 1;
  int top = 31;
  while (!((value >> top) & 1u)) --top;
  uint32_t c = 0;
  unsigned n = 0;
  for (int b = top - 1; b >= 0; --b) { c = (c << 2) | ((value >> b) & 1u); n += 2; }
  c = (c << 1) | 1u;
  *bits = c;
  return n + 1;
}
/* VLC.cpp:78-85 */
static unsigned svlc_code(int32_t value, uint32_t *bits) {
  if (value == 0) { *bits = 1; return 1; }
  const uint32_t mag = value < 0 ? 0u - (uint32_t)value : (uint32_t)value;
  unsigned n = uvlc_code(mag, bits);
  *bits = (*bits << 1) | (value < 0 ? 1u : 0u);
  return n + 1;
}
static void put_uvlc(bitw *w, uint32_t v) { uint32_t c; unsigned n = uvlc_code(v, &c); put_bits(w, n, c); }
static void put_svlc(bitw *w, int32_t v) { uint32_t c; unsigned n = svlc_code(v, &c); put_bits(w, n, c); }

typedef struct {
  const uint8_t *buf;
  size_t len, pos;
  unsigned cache;
  int cached, bounded, eof;
  long left;
} bitr;

/* VLC.cpp:182-202 */
static int get_bit(bitr *r) {
  if (r->bounded && r->left < 1) return 1;
  if (r->cached == 0) {
    if (r->pos < r->len) r->cache = r->buf[r->pos]; else { r->cache = 0xFF; r->eof = 1; }
    ++r->pos;
    r->cached = 8;
  }
  --r->cached;
  --r->left;
  return (r->cache >> r->cached) & 1u;
}
static uint32_t get_bits(bitr *r, unsigned n) { uint32_t v = 0; while (n--) v = (v << 1) | (uint32_t)get_bit(r); return v; }
static void r_bounded(bitr *r, long bits) { r->bounded = 1; r->left = bits; }
static void r_flush(bitr *r) { if (r->bounded) while (r->left > 0) get_bit(r); }
static void r_align(bitr *r) { r->bounded = 0; while (r->cached) get_bit(r); }
static uint32_t get_bytes(bitr *r, int n) { /* :337-348 */
  r_align(r);
  uint32_t v = 0;
  while (n-- > 0) {
    uint32_t b = 0xFF;
    if (r->pos < r->len) b = r->buf[r->pos]; else r->eof = 1;
    ++r->pos;
    v = (v << 8) | b;
  }
  return v;
}
/* VLC.cpp:283-295 + :54-66 */
static uint32_t get_uvlc(bitr *r) {
  uint32_t value = 1;
  while (!get_bit(r)) value = (value << 1) | (uint32_t)get_bit(r);
  return value - 1;
}
/* VLC.cpp:304-317 + :87-94 */
static int32_t get_svlc(bitr *r) {
  const uint32_t mag = get_uvlc(r);
  if (mag == 0) return 0;
  return get_bit(r) ? (int32_t)(0u - mag) : (int32_t)mag;
}

/* ------------------------------------------------------------------------ */
/* slices                                                                    */
/* ------------------------------------------------------------------------ */

static int gcd_i(int a, int b) { a = abs(a); b = abs(b); while (b) { int t = a % b; a = b; b = t; } return a; }

/* Slices.cpp:28-49 */
int vc2o_slice_bytes(int ys, int xs, int total_bytes, int scalar, int32_t *out) {
  const int n = ys * xs;
  int num = total_bytes / scalar - 4 * n, den = n;
  const int g = gcd_i(num, den);
  if (g) { num /= g; den /= g; }
  const int ratio = num / den, remainder = num - ratio * den;
  int residue = 0;
  for (int i = 0; i < n; ++i) {
    residue += remainder;
    if (residue < den) out[i] = ratio * scalar + 4;
    else { out[i] = (ratio + 1) * scalar + 4; residue -= den; }
  }
  return 0;
}

/* one slice of one component: tile origin + size inside a padded plane */
typedef struct { const int32_t *p; int pw, y0, x0, sh, sw; } tile;
static tile tile_of(const int32_t *plane, int ph, int pw, int ys, int xs, int v, int h) {
  tile t = {plane, pw, v * (ph / ys), h * (pw / xs), ph / ys, pw / xs};
  return t;
}
#define TILE_AT(t, y, x) ((t).p[(size_t)((t).y0 + (y)) * (t).pw + (size_t)((t).x0 + (x))])

/* Slices.cpp:97-119 core: bits through the last non-zero coefficient, coding order */
static int tile_count_bits(const tile *t, int depth) {
  int count = 0, gross = 0;
  for (int band = 0; band < 3 * depth + 1; ++band) {
    int s, oy, ox;
    band_geom(band, depth, &s, &oy, &ox);
    for (int y = oy; y < t->sh; y += s)
      for (int x = ox; x < t->sw; x += s) {
        uint32_t c;
        const int nb = (int)svlc_code(TILE_AT(*t, y, x), &c);
        gross += nb;
        if (nb > 1) count = gross;
      }
  }
  return count;
}
static int tile_bytes(const tile *t, int depth, int scalar, int *out) {
  const int count = tile_count_bits(t, depth);
  const int len = ((count + 7) / 8 + scalar - 1) / scalar;
  if (len > 0xFF) return fail(VC2O_ESCALAR, "Slice scalar is too small, consider using a larger slice scalar.");
  *out = len * scalar;
  return 0;
}
int vc2o_component_slice_bytes(const int32_t *plane, int ph, int pw, int depth, int ys, int xs, int v,
                               int h, int scalar, int32_t *bytes_out) {
  const tile t = tile_of(plane, ph, pw, ys, xs, v, h);
  return tile_bytes(&t, depth, scalar, bytes_out);
}

static void write_tile(bitw *w, const tile *t, int depth) {
  for (int band = 0; band < 3 * depth + 1; ++band) {
    int s, oy, ox;
    band_geom(band, depth, &s, &oy, &ox);
    for (int y = oy; y < t->sh; y += s)
      for (int x = ox; x < t->sw; x += s) put_svlc(w, TILE_AT(*t, y, x));
  }
}
static void read_tile(bitr *r, int32_t *plane, const tile *t, int depth) {
  for (int band = 0; band < 3 * depth + 1; ++band) {
    int s, oy, ox;
    band_geom(band, depth, &s, &oy, &ox);
    for (int y = oy; y < t->sh; y += s)
      for (int x = ox; x < t->sw; x += s)
        plane[(size_t)(t->y0 + y) * t->pw + (size_t)(t->x0 + x)] = get_svlc(r);
  }
}

/* Slices.cpp:469-533 (VBR) and :305-382 (CBR), over Slices.cpp:645-660 */
int vc2o_hq_pack(const int32_t *y, const int32_t *u, const int32_t *v, const vc2o_geom *g,
                 const int32_t *qidx, int prefix, int scalar, const int32_t *cbr, uint8_t *out,
                 size_t cap, size_t *out_len) {
  bitw w = {out, cap, 0, 0, 0, 0, 0, 0};
  const int32_t *planes[3] = {y, u, v};
  for (int sv = 0; sv < g->y_slices; ++sv)
    for (int sh = 0; sh < g->x_slices; ++sh) {
      const int si = sv * g->x_slices + sh;
      for (int n = 0; n < prefix; ++n) put_bytes(&w, 1, 0);
      put_bytes(&w, 1, (uint32_t)qidx[si] & 0xFF);
      int used = 0;
      for (int c = 0; c < 3; ++c) {
        const int ph = c ? g->chroma_h : g->luma_h, pw = c ? g->chroma_w : g->luma_w;
        const tile t = tile_of(planes[c], ph, pw, g->y_slices, g->x_slices, sv, sh);
        int bytes;
        int rc = tile_bytes(&t, g->depth, scalar, &bytes);
        if (rc) return rc;
        if (cbr && c == 2) {
          const int vbytes = cbr[si] - 4 - used;
          if (vbytes < bytes) return fail(VC2O_ECBR_TOOBIG, "SliceIO, HQ CBR mode: Too many bytes for the slice");
          if (vbytes / scalar > 255)
            return fail(VC2O_ECBR_LEN, "Slice component length exceeds 1 byte when divided by slice size scalar. See above for suggestions to prevent this.");
          bytes = vbytes;
        }
        used += bytes;
        put_bytes(&w, 1, (uint32_t)(bytes / scalar));
        w_bounded(&w, 8L * bytes);
        write_tile(&w, &t, g->depth);
        w_flush(&w);
        w_align(&w);
        if (w.err == VC2O_EBOUNDED) return fail(VC2O_EBOUNDED, "Attempt to write beyond end of bounded write");
      }
    }
  if (w.err) return fail(w.err, "output buffer too small");
  *out_len = w.pos;
  return 0;
}

/* Slices.cpp:535-612 over :662-694 (DecodeStream always parses HQ as VBR) */
int vc2o_hq_unpack(const uint8_t *in, size_t len, const vc2o_geom *g, int prefix, int scalar,
                   int32_t *y, int32_t *u, int32_t *v, int32_t *qidx, size_t *consumed) {
  bitr r = {in, len, 0, 0, 0, 0, 0, 0};
  int32_t *planes[3] = {y, u, v};
  for (int sv = 0; sv < g->y_slices; ++sv)
    for (int sh = 0; sh < g->x_slices; ++sh) {
      for (int n = 0; n < prefix; ++n) get_bytes(&r, 1);
      qidx[sv * g->x_slices + sh] = (int32_t)get_bytes(&r, 1);
      for (int c = 0; c < 3; ++c) {
        const int ph = c ? g->chroma_h : g->luma_h, pw = c ? g->chroma_w : g->luma_w;
        const tile t = tile_of(planes[c], ph, pw, g->y_slices, g->x_slices, sv, sh);
        const int bytes = (int)get_bytes(&r, 1) * scalar;
        r_bounded(&r, 8L * bytes);
        read_tile(&r, planes[c], &t, g->depth);
        r_flush(&r);
        r_align(&r);
      }
    }
  if (r.eof) return fail(VC2O_ESTREAM, "Failed to read HQ compressed frame");
  if (consumed) *consumed = r.pos;
  return 0;
}

/* quantise one slice tile of a transform plane with a single index (Quantisation.cpp:507-519) */
static int quantise_tile(const tile *t, int depth, int q, const int32_t *qm, int32_t *dst /* sh*sw */) {
  for (int yy = 0; yy < t->sh; ++yy)
    for (int xx = 0; xx < t->sw; ++xx) {
      const int aq = adjust_q(q, qm[band_of(yy, xx, depth)]);
      const int rc = vc2o_quant(TILE_AT(*t, yy, xx), aq, &dst[yy * t->sw + xx]);
      if (rc) return rc;
    }
  return 0;
}

/* Quantisation.cpp:627-642 (luma only; squared difference in int, sum in long long) */
static int yss_tile(const tile *t, int depth, int q, const int32_t *qm, long long *out) {
  long long acc = 0;
  for (int yy = 0; yy < t->sh; ++yy)
    for (int xx = 0; xx < t->sw; ++xx) {
      const int aq = adjust_q(q, qm[band_of(yy, xx, depth)]);
      int32_t qv, rv;
      int rc = vc2o_quant(TILE_AT(*t, yy, xx), aq, &qv);
      if (rc) return rc;
      if ((rc = vc2o_scale(qv, aq, &rv))) return rc;
      const int32_t d = (int32_t)((uint32_t)TILE_AT(*t, yy, xx) - (uint32_t)rv);
      acc += (int32_t)((uint32_t)d * (uint32_t)d);
    }
  *out = acc;
  return 0;
}

/* EncodeStream.cpp:73-125 */
int vc2o_cbr_qindices(const int32_t *y, const int32_t *u, const int32_t *v, const vc2o_geom *g,
                      const int32_t *qm, const int32_t *slice_bytes, int scalar, int32_t *qidx) {
  const int32_t *planes[3] = {y, u, v};
  const int lsh = g->luma_h / g->y_slices, lsw = g->luma_w / g->x_slices;
  int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)lsh * lsw);
  int rc = 0;
  for (int sv = 0; sv < g->y_slices && !rc; ++sv)
    for (int sh = 0; sh < g->x_slices && !rc; ++sh) {
      const int si = sv * g->x_slices + sh;
      const int avail = slice_bytes[si] - 4;
      int trial = 63, q = 127, delta = 64;
      while (delta > 0 && !rc) {
        delta >>= 1;
        int need = 0;
        for (int c = 0; c < 3 && !rc; ++c) {
          const int ph = c ? g->chroma_h : g->luma_h, pw = c ? g->chroma_w : g->luma_w;
          const tile t = tile_of(planes[c], ph, pw, g->y_slices, g->x_slices, sv, sh);
          if ((rc = quantise_tile(&t, g->depth, trial, qm, tmp))) break;
          const tile qt = {tmp, t.sw, 0, 0, t.sh, t.sw};
          int b;
          if ((rc = tile_bytes(&qt, g->depth, scalar, &b))) break;
          need += b;
        }
        if (rc) break;
        if (need <= avail) { if (trial < q) q = trial; trial -= delta; }
        else trial += delta;
      }
      if (rc) break;
      const tile lt = tile_of(y, g->luma_h, g->luma_w, g->y_slices, g->x_slices, sv, sh);
      trial = q;
      long long prev, cur, d;
      if ((rc = yss_tile(&lt, g->depth, trial, qm, &prev))) break;
      do {
        ++trial;
        if ((rc = yss_tile(&lt, g->depth, trial, qm, &cur))) break;
        d = cur - prev;
        prev = cur;
      } while (d < 0);
      if (rc) break;
      qidx[si] = trial - 1;
    }
  free(tmp);
  return rc;
}

/* Utils.cpp:40-48 */
static int intlog2(int value) { int l = 0; --value; while (value > 0) { value >>= 1; ++l; } return l; }

/* Slices.cpp:51-69 / :71-95 */
static int luma_bits(const tile *t, int depth) { return tile_count_bits(t, depth); }
static int chroma_bits(const tile *tu, const tile *tv, int depth) {
  int count = 0, gross = 0;
  for (int band = 0; band < 3 * depth + 1; ++band) {
    int s, oy, ox;
    band_geom(band, depth, &s, &oy, &ox);
    for (int y = oy; y < tu->sh; y += s)
      for (int x = ox; x < tu->sw; x += s) {
        uint32_t c;
        int nb = (int)svlc_code(TILE_AT(*tu, y, x), &c);
        gross += nb; if (nb > 1) count = gross;
        nb = (int)svlc_code(TILE_AT(*tv, y, x), &c);
        gross += nb; if (nb > 1) count = gross;
      }
  }
  return count;
}

/* Slices.cpp:195-244 over :645-660 */
int vc2o_ld_pack(const int32_t *y, const int32_t *u, const int32_t *v, const vc2o_geom *g,
                 const int32_t *qidx, const int32_t *slice_bytes, uint8_t *out, size_t cap,
                 size_t *out_len) {
  bitw w = {out, cap, 0, 0, 0, 0, 0, 0};
  for (int sv = 0; sv < g->y_slices; ++sv)
    for (int sh = 0; sh < g->x_slices; ++sh) {
      const int si = sv * g->x_slices + sh, size = slice_bytes[si];
      const tile ty = tile_of(y, g->luma_h, g->luma_w, g->y_slices, g->x_slices, sv, sh);
      const tile tu = tile_of(u, g->chroma_h, g->chroma_w, g->y_slices, g->x_slices, sv, sh);
      const tile tv = tile_of(v, g->chroma_h, g->chroma_w, g->y_slices, g->x_slices, sv, sh);
      put_bits(&w, 7, (uint32_t)qidx[si]);
      const int ybits = luma_bits(&ty, g->depth);
      const int split = intlog2(8 * size - 7);
      const int uvbits = 8 * size - 7 - split - ybits;
      if (uvbits < chroma_bits(&tu, &tv, g->depth))
        return fail(VC2O_ELD_TOOBIG, "SliceIO, LD mode: Too many bytes for the U and V slices");
      put_bits(&w, (unsigned)split, (uint32_t)ybits);
      w_bounded(&w, ybits);
      write_tile(&w, &ty, g->depth);
      w_flush(&w);
      w_bounded(&w, uvbits);
      for (int band = 0; band < 3 * g->depth + 1; ++band) {
        int s, oy, ox;
        band_geom(band, g->depth, &s, &oy, &ox);
        for (int yy = oy; yy < tu.sh; yy += s)
          for (int xx = ox; xx < tu.sw; xx += s) {
            put_svlc(&w, TILE_AT(tu, yy, xx));
            put_svlc(&w, TILE_AT(tv, yy, xx));
          }
      }
      w_flush(&w);
      w_align(&w);
      if (w.err == VC2O_EBOUNDED) return fail(VC2O_EBOUNDED, "Attempt to write beyond end of bounded write");
    }
  if (w.err) return fail(w.err, "output buffer too small");
  *out_len = w.pos;
  return 0;
}

/* Slices.cpp:246-303 over :662-694 */
int vc2o_ld_unpack(const uint8_t *in, size_t len, const vc2o_geom *g, const int32_t *slice_bytes,
                   int32_t *y, int32_t *u, int32_t *v, int32_t *qidx, size_t *consumed) {
  bitr r = {in, len, 0, 0, 0, 0, 0, 0};
  for (int sv = 0; sv < g->y_slices; ++sv)
    for (int sh = 0; sh < g->x_slices; ++sh) {
      const int si = sv * g->x_slices + sh, size = slice_bytes[si];
      const tile ty = tile_of(y, g->luma_h, g->luma_w, g->y_slices, g->x_slices, sv, sh);
      const tile tu = tile_of(u, g->chroma_h, g->chroma_w, g->y_slices, g->x_slices, sv, sh);
      qidx[si] = (int32_t)get_bits(&r, 7);
      const int split = intlog2(8 * size - 7);
      const int ybits = (int)get_bits(&r, (unsigned)split);
      const int uvbits = 8 * size - 7 - split - ybits;
      r_bounded(&r, ybits);
      read_tile(&r, y, &ty, g->depth);
      r_flush(&r);
      r_bounded(&r, uvbits);
      for (int band = 0; band < 3 * g->depth + 1; ++band) {
        int s, oy, ox;
        band_geom(band, g->depth, &s, &oy, &ox);
        for (int yy = oy; yy < tu.sh; yy += s)
          for (int xx = ox; xx < tu.sw; xx += s) {
            const size_t i = (size_t)(tu.y0 + yy) * tu.pw + (size_t)(tu.x0 + xx);
            u[i] = get_svlc(&r);
            v[i] = get_svlc(&r);
          }
      }
      r_flush(&r);
      r_align(&r);
    }
  if (r.eof) return fail(VC2O_ESTREAM, "Failed to read LD compressed frame");
  if (consumed) *consumed = r.pos;
  return 0;
}

/* EncodeStream.cpp:141-245: per-slice search with the DC-prediction state machine */
typedef struct { const int32_t *coef; int ph, pw, lw; int32_t *decoded_ll; } ldq_plane;
static int ldq_slice(ldq_plane *p, const vc2o_geom *g, int sv, int sh, int q, const int32_t *qm,
                     int32_t *dst) {
  const tile t = tile_of(p->coef, p->ph, p->pw, g->y_slices, g->x_slices, sv, sh);
  const int ts = 1 << g->depth;
  for (int yy = 0; yy < t.sh; ++yy)
    for (int xx = 0; xx < t.sw; ++xx) {
      const int aq = adjust_q(q, qm[band_of(yy, xx, g->depth)]);
      int rc;
      if (yy % ts == 0 && xx % ts == 0) {
        const int yl = (t.y0 + yy) / ts, xl = (t.x0 + xx) / ts;
        const int32_t pred = predict_dc(p->decoded_ll, p->lw, yl, xl);
        int32_t qv, rv;
        if ((rc = vc2o_quant(TILE_AT(t, yy, xx) - pred, aq, &qv))) return rc;
        if ((rc = vc2o_scale(qv, aq, &rv))) return rc;
        dst[yy * t.sw + xx] = qv;
        p->decoded_ll[yl * p->lw + xl] = rv + pred;
      } else if ((rc = vc2o_quant(TILE_AT(t, yy, xx), aq, &dst[yy * t.sw + xx]))) return rc;
    }
  return 0;
}
int vc2o_ld_qindices(const int32_t *y, const int32_t *u, const int32_t *v, const vc2o_geom *g,
                     const int32_t *qm, const int32_t *slice_bytes, int32_t *qidx) {
  const int ts = 1 << g->depth;
  ldq_plane pl[3] = {{y, g->luma_h, g->luma_w, g->luma_w / ts, NULL},
                     {u, g->chroma_h, g->chroma_w, g->chroma_w / ts, NULL},
                     {v, g->chroma_h, g->chroma_w, g->chroma_w / ts, NULL}};
  int32_t *tmp[3];
  int rc = 0;
  for (int c = 0; c < 3; ++c) {
    pl[c].decoded_ll = (int32_t *)calloc((size_t)(pl[c].ph / ts) * pl[c].lw, sizeof(int32_t));
    tmp[c] = (int32_t *)malloc(sizeof(int32_t) * (size_t)(pl[c].ph / g->y_slices) * (pl[c].pw / g->x_slices));
  }
  for (int sv = 0; sv < g->y_slices && !rc; ++sv)
    for (int sh = 0; sh < g->x_slices && !rc; ++sh) {
      const int si = sv * g->x_slices + sh, bytes = slice_bytes[si];
      const int avail = 8 * bytes - 7 - intlog2(8 * bytes - 7);
      int trial = 63, q = 127, delta = 64;
      while (delta > 0 && !rc) {
        delta >>= 1;
        for (int c = 0; c < 3 && !rc; ++c) rc = ldq_slice(&pl[c], g, sv, sh, trial, qm, tmp[c]);
        if (rc) break;
        const tile ty = {tmp[0], pl[0].pw / g->x_slices, 0, 0, pl[0].ph / g->y_slices, pl[0].pw / g->x_slices};
        const tile tu = {tmp[1], pl[1].pw / g->x_slices, 0, 0, pl[1].ph / g->y_slices, pl[1].pw / g->x_slices};
        const tile tv = {tmp[2], tu.pw, 0, 0, tu.sh, tu.sw};
        const int need = luma_bits(&ty, g->depth) + chroma_bits(&tu, &tv, g->depth);
        if (need <= avail) { if (trial < q) q = trial; trial -= delta; }
        else trial += delta;
      }
      for (int c = 0; c < 3 && !rc; ++c) rc = ldq_slice(&pl[c], g, sv, sh, q, qm, tmp[c]);
      qidx[si] = q;
    }
  for (int c = 0; c < 3; ++c) { free(pl[c].decoded_ll); free(tmp[c]); }
  return rc;
}

/* ------------------------------------------------------------------------ */
/* stream syntax                                                             */
/* ------------------------------------------------------------------------ */

/* DataUnit.cpp:80-123 */
size_t vc2o_write_parse_info(uint8_t *out, int parse_code, uint32_t next, uint32_t prev) {
  out[0] = 0x42; out[1] = 0x42; out[2] = 0x43; out[3] = 0x44;
  out[4] = (uint8_t)parse_code;
  for (int i = 0; i < 4; ++i) { out[5 + i] = (uint8_t)(next >> (24 - 8 * i)); out[9 + i] = (uint8_t)(prev >> (24 - 8 * i)); }
  return 13;
}

/* DataUnit.cpp:430-459 getDefaultSourceParameters as data:
 * {height,width,cf,interlace,frame_rate,tff,bitdepth,aspect,clean_w,clean_h,left,top,colour_spec} */
static const int BASE_FORMATS[23][13] = {
    {480, 640, 2, 0, 1, 0, 8, 1, 640, 480, 0, 0, 0},     {120, 176, 2, 0, 9, 0, 8, 2, 176, 120, 0, 0, 1},
    {144, 176, 2, 0, 10, 1, 8, 3, 176, 144, 0, 0, 2},    {240, 352, 2, 0, 9, 0, 8, 2, 352, 240, 0, 0, 1},
    {288, 352, 2, 0, 10, 1, 8, 3, 352, 288, 0, 0, 2},    {480, 704, 2, 0, 9, 0, 8, 2, 704, 480, 0, 0, 1},
    {576, 704, 2, 0, 10, 1, 8, 3, 704, 576, 0, 0, 2},    {480, 720, 1, 1, 4, 0, 10, 2, 704, 480, 8, 0, 1},
    {576, 720, 1, 1, 3, 1, 10, 3, 704, 576, 8, 0, 2},    {720, 1280, 1, 0, 7, 1, 10, 1, 1280, 720, 0, 0, 3},
    {720, 1280, 1, 0, 6, 1, 10, 1, 1280, 720, 0, 0, 3},  {1080, 1920, 1, 1, 4, 1, 10, 1, 1920, 1080, 0, 0, 3},
    {1080, 1920, 1, 1, 3, 1, 10, 1, 1920, 1080, 0, 0, 3}, {1080, 1920, 1, 0, 7, 1, 10, 1, 1920, 1080, 0, 0, 3},
    {1080, 1920, 1, 0, 6, 1, 10, 1, 1920, 1080, 0, 0, 3}, {1080, 2048, 0, 0, 2, 1, 12, 1, 2048, 1080, 0, 0, 4},
    {2160, 4096, 0, 0, 2, 1, 12, 1, 4096, 2160, 0, 0, 4}, {2160, 3840, 1, 0, 7, 1, 10, 1, 3840, 2160, 0, 0, 5},
    {2160, 3840, 1, 0, 6, 1, 10, 1, 3840, 2160, 0, 0, 5}, {4320, 7680, 1, 0, 7, 1, 10, 1, 7680, 4320, 0, 0, 5},
    {4320, 7680, 1, 0, 6, 1, 10, 1, 7680, 4320, 0, 0, 5}, {1080, 1920, 1, 0, 1, 1, 10, 1, 1920, 1080, 0, 0, 3},
    {486, 720, 1, 1, 4, 0, 10, 2, 720, 486, 0, 0, 3}};
enum { BF_H, BF_W, BF_CF, BF_IL, BF_FR, BF_TFF, BF_BD };

typedef struct { int width, height, cf, interlace, fr, tff, bd; } seq_fmt;

/* DataUnit.cpp:486-503: all-field match (optional fields are unset by EncodeStream) */
static int matches_index(const seq_fmt *f, int i) {
  const int *b = BASE_FORMATS[i];
  return f->width == b[BF_W] && f->height == b[BF_H] && f->cf == b[BF_CF] && f->fr == b[BF_FR] &&
         f->bd == b[BF_BD] && f->interlace == b[BF_IL] && f->tff == b[BF_TFF];
}
/* DataUnit.cpp:470-484 */
static int matches_explicit(const seq_fmt *f, int w, int h, int cf, int fr, int bd, int tff) {
  return f->width == w && f->height == h && f->cf == cf && f->fr == fr && f->bd == bd && f->tff == tff;
}
/* DataUnit.cpp:505-530 */
static int check_match(const seq_fmt *f, int i) {
  const int *b = BASE_FORMATS[i];
  const int n = (f->width != b[BF_W]) + (f->height != b[BF_H]) + (f->cf != b[BF_CF]) +
                (f->fr != b[BF_FR]) + (f->bd != b[BF_BD]) + (f->interlace != b[BF_IL]);
  return f->tff == b[BF_TFF] ? n : -1;
}

/* DataUnit.cpp:563-784 (video_format from SequenceHeader) + :786-881 (writer),
 * for the SequenceHeader EncodeStream.cpp:443-450 builds: optional fields unset. */
int vc2o_write_sequence_header_payload(const vc2o_params *p, uint8_t *out, size_t cap, size_t *len,
                                       int *major_version_out) {
  const seq_fmt f = {p->width, p->height, p->cf, p->interlaced ? 1 : 0, p->frame_rate, p->bottom_field_first ? 0 : 1, p->bit_depth};
  const int profile_hq = (p->mode != 2);
  int major = profile_hq ? 2 : 1; /* DataUnit.cpp:417-426 */
  if (f.fr > 11 /* MAX_V2_FRAMERATE = FR48 */ || f.bd > 12) major = 3;
  if (p->fragment_length > 0 && p->mode != 0 && major < 3) major = 3; /* DataUnit.cpp:1065-1067 */
  int base = 0, level = 0;
  int custom_dims = 0, custom_cf = 0, custom_scan = 0, source_sampling = 0, custom_fr = 0, fr = 0;
  int custom_clean = 0, custom_range = 0, range_index = 0;
  int fw = 0, fh = 0, cfv = 0;

  if (f.interlace) { /* DataUnit.cpp:612-632 */
    if (matches_index(&f, 7)) { base = 7; level = 2; }
    else if (matches_index(&f, 8)) { base = 8; level = 2; }
    else if (matches_index(&f, 22)) { base = 22; level = 2; }
    else if (f.cf == 1 && f.width == 720 && f.height >= 480 && f.height <= 486 && f.fr == 4 && f.bd == 10) {
      base = 7; level = 2; custom_dims = 1; fw = f.width; fh = f.height;
    }
    else if (matches_index(&f, 11)) { base = 11; level = 3; }
    else if (matches_index(&f, 12)) { base = 12; level = 3; }
  }
  /* progressive branch of DataUnit.cpp:633-675 */
  static const int simple[] = {1, 2, 3, 4, 5, 6};
  for (unsigned i = 0; i < 6 && !base && !f.interlace; ++i)
    if (matches_index(&f, simple[i])) { base = simple[i]; level = 1; }
  if (!base && !f.interlace) {
    if (matches_explicit(&f, 720, 480, 1, 4, 10, 0)) { base = 7; level = 2; custom_scan = 1; }
    else if (matches_explicit(&f, 720, 576, 1, 3, 10, 1)) { base = 8; level = 2; custom_scan = 1; }
    else if (matches_explicit(&f, 720, 486, 1, 4, 10, 0)) { base = 22; level = 2; custom_scan = 1; }
    else if (matches_index(&f, 9)) { base = 9; level = 3; }
    else if (matches_index(&f, 10)) { base = 10; level = 3; }
    else if (matches_explicit(&f, 1920, 1080, 1, 4, 10, 1)) { base = 11; level = 3; custom_scan = 1; }
    else if (matches_explicit(&f, 1920, 1080, 1, 3, 10, 1)) { base = 12; level = 3; custom_scan = 1; }
    else if (matches_index(&f, 13)) { base = 13; level = 3; }
    else if (matches_index(&f, 14)) { base = 14; level = 3; }
    else if (matches_index(&f, 21)) { base = 21; level = 3; }
    else if (matches_index(&f, 15)) { base = 15; level = 4; }
    else if (matches_explicit(&f, 2048, 1080, 0, 11, 12, 1)) { base = 15; level = 4; custom_fr = 1; fr = 11; }
    else if (matches_index(&f, 16)) { base = 16; level = 5; }
    else if (matches_index(&f, 17)) { base = 17; level = 6; }
    else if (matches_index(&f, 18)) { base = 18; level = 6; }
    else if (matches_index(&f, 19)) { base = 19; level = 7; }
    else if (matches_index(&f, 20)) { base = 20; level = 7; }
  }
  if (!base) { /* DataUnit.cpp:677-783: closest base format + custom flags */
    level = 0;
    custom_dims = 0;
    int best = 999;
    for (int i = 1; i <= 22; ++i) {
      const int n = check_match(&f, i);
      if (n == -1) continue;
      if (n < best) { base = i; best = n; }
    }
    const int *b = BASE_FORMATS[base];
    if (f.interlace != b[BF_IL]) { custom_scan = 1; source_sampling = f.interlace; }
    if (f.width != b[BF_W] || f.height != b[BF_H]) { custom_dims = 1; fw = f.width; fh = f.height; }
    if (f.cf != b[BF_CF]) { custom_cf = 1; cfv = f.cf; }
    if (f.fr != b[BF_FR]) { custom_fr = 1; fr = f.fr; }
    if (f.bd != b[BF_BD]) {
      custom_range = 1;
      switch (f.bd) {
        case 8: range_index = 1; break;
        case 10: range_index = 3; break;
        case 12: range_index = 4; break;
        case 16: range_index = 7; break;
        default: return fail(VC2O_EINVAL, "DataUnitIO: invalid bit depth");
      }
    }
    if (custom_dims) custom_clean = 1; /* clean area = full picture, :745-756 */
  }

  bitw w = {out, cap, 0, 0, 0, 0, 0, 0};
  put_uvlc(&w, (uint32_t)major);
  put_uvlc(&w, 0);
  put_uvlc(&w, profile_hq ? 3u : 0u);
  put_uvlc(&w, (uint32_t)level);
  put_uvlc(&w, (uint32_t)base);
  put_bit(&w, custom_dims);
  if (custom_dims) { put_uvlc(&w, (uint32_t)fw); put_uvlc(&w, (uint32_t)fh); }
  put_bit(&w, custom_cf);
  if (custom_cf) put_uvlc(&w, (uint32_t)cfv);
  put_bit(&w, custom_scan);
  if (custom_scan) put_uvlc(&w, (uint32_t)source_sampling);
  put_bit(&w, custom_fr);
  if (custom_fr) put_uvlc(&w, (uint32_t)fr);
  put_bit(&w, 0); /* pixel aspect ratio */
  put_bit(&w, custom_clean);
  if (custom_clean) { put_uvlc(&w, (uint32_t)fw); put_uvlc(&w, (uint32_t)fh); put_uvlc(&w, 0); put_uvlc(&w, 0); }
  put_bit(&w, custom_range);
  if (custom_range) put_uvlc(&w, (uint32_t)range_index);
  put_bit(&w, 0); /* colour spec */
  put_uvlc(&w, (uint32_t)source_sampling); /* picture coding mode, :873-877 */
  w_align(&w);
  if (w.err) return fail(w.err, "output buffer too small");
  *len = w.pos;
  *major_version_out = major;
  return 0;
}

/* DataUnit.cpp:241-259 */
int vc2o_write_hq_picture_header(uint32_t picture_number, int kernel, int depth, int slices_x,
                                 int slices_y, int prefix, int scalar, int major_version,
                                 uint8_t *out, size_t cap, size_t *len) {
  bitw w = {out, cap, 0, 0, 0, 0, 0, 0};
  put_bytes(&w, 4, picture_number);
  put_uvlc(&w, (uint32_t)kernel);
  put_uvlc(&w, (uint32_t)depth);
  if (major_version >= 3) { put_bit(&w, 0); put_bit(&w, 0); }
  put_uvlc(&w, (uint32_t)slices_x);
  put_uvlc(&w, (uint32_t)slices_y);
  put_uvlc(&w, (uint32_t)prefix);
  put_uvlc(&w, (uint32_t)scalar);
  put_bit(&w, 0);
  w_align(&w);
  if (w.err) return fail(w.err, "output buffer too small");
  *len = w.pos;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* whole-file drivers                                                        */
/* ------------------------------------------------------------------------ */

static int make_geom(const vc2o_params *p, vc2o_geom *g, int decoder_side) {
  int ch, cw;
  vc2o_chroma_dims(p->height, p->width, p->cf, &ch, &cw);
  g->depth = p->depth;
  g->luma_h = vc2o_padded_size(p->height, p->depth);
  g->luma_w = vc2o_padded_size(p->width, p->depth);
  if (decoder_side) /* DecodeStream.cpp:483-498: chroma derived from the PADDED luma size */
    vc2o_chroma_dims(g->luma_h, g->luma_w, p->cf, &g->chroma_h, &g->chroma_w);
  else { /* WaveletTransform.cpp:1267-1273 */
    g->chroma_h = vc2o_padded_size(ch, p->depth);
    g->chroma_w = vc2o_padded_size(cw, p->depth);
  }
  return 0;
}

/* Frame.cpp:40-88: the rows of one field of a plane (top = even rows) */
static void take_field(const int32_t *plane, int h, int w, int first, int32_t *field) {
  for (int y = first, r = 0; y < h; y += 2, ++r) memcpy(field + (size_t)r * w, plane + (size_t)y * w, sizeof(int32_t) * w);
}

/* bytes of one HQ slice as serialised (Slices.cpp:305-382 / :469-533): prefix, index, 3 x (length byte + data) */
static size_t hq_slice_size(const uint8_t *s, size_t avail, int prefix, int scalar) {
  size_t q = (size_t)prefix + 1;
  for (int c = 0; c < 3; ++c) {
    if (q >= avail) return 0;
    q += 1 + (size_t)s[q] * (size_t)scalar;
  }
  return q <= avail ? q : 0;
}

static void put_be(uint8_t *d, int n, uint32_t v) { for (int i = 0; i < n; ++i) d[i] = (uint8_t)(v >> (8 * (n - 1 - i))); }

/* EncodeStream.cpp:247-788, -o Stream (progressive or interlaced, whole pictures or fragments) */
int vc2o_encode_stream(const vc2o_params *p, const uint8_t *raw, int n_frames, uint8_t *out,
                       size_t cap, size_t *out_len) {
  const int frame_pics = p->interlaced ? 2 : 1; /* EncodeStream.cpp:431 */
  vc2o_params pp = *p;                           /* the coded picture: a frame or one field, :367-368 */
  if (p->interlaced) pp.height = p->height / 2;
  vc2o_geom g;
  make_geom(&pp, &g, 0);
  int ch, cw, fch, fcw;
  vc2o_chroma_dims(pp.height, pp.width, pp.cf, &ch, &cw);    /* picture chroma */
  vc2o_chroma_dims(p->height, p->width, p->cf, &fch, &fcw);  /* frame chroma */
  g.y_slices = vc2o_slice_size_is_valid(p->depth, pp.height, ch, p->y_size);
  g.x_slices = vc2o_slice_size_is_valid(p->depth, pp.width, cw, p->x_size);
  if (!g.y_slices || !g.x_slices)
    return fail(VC2O_EINVAL, "The given waveletDepth, hSlice, and vSlice parameters cannot encode this input. See above for suggested parameters.");
  const int n_slices = g.y_slices * g.x_slices;
  const int picture_bytes = p->interlaced ? p->compressed_bytes / 2 : p->compressed_bytes; /* :378 */
  const int fragmented = p->fragment_length > 0 && p->mode != 0;                            /* :444-445 */
  int32_t qm[3 * 31 + 1];
  int rc = vc2o_quant_matrix(p->kernel, p->depth, qm);
  if (rc) return rc;

  const size_t fln = (size_t)p->height * p->width, fcn = (size_t)fch * fcw;
  const size_t pln = (size_t)g.luma_h * g.luma_w, pcn = (size_t)g.chroma_h * g.chroma_w;
  const size_t frame_bytes = (fln + 2 * fcn) * p->word_bytes;
  int32_t *in = (int32_t *)malloc(sizeof(int32_t) * (fln > fcn ? fln : fcn));
  int32_t *fld = (int32_t *)malloc(sizeof(int32_t) * (fln > fcn ? fln : fcn));
  int32_t *tr[3], *qc[3];
  for (int c = 0; c < 3; ++c) {
    tr[c] = (int32_t *)malloc(sizeof(int32_t) * (c ? pcn : pln));
    qc[c] = (int32_t *)malloc(sizeof(int32_t) * (c ? pcn : pln));
  }
  int32_t *qidx = (int32_t *)malloc(sizeof(int32_t) * n_slices);
  int32_t *sbytes = (int32_t *)malloc(sizeof(int32_t) * n_slices);
  uint8_t *slices = NULL; /* fragmented pictures: the slice bytes before they are cut into fragments */
  size_t slices_cap = 0;

  size_t pos = 0;
  uint32_t prev = 0;
  int major = 0;
  { /* sequence header data unit, DataUnit.cpp:1062-1078 */
    uint8_t payload[256];
    size_t plen;
    if ((rc = vc2o_write_sequence_header_payload(p, payload, sizeof payload, &plen, &major))) goto done;
    if (pos + 13 + plen > cap) { rc = fail(VC2O_ECAP, "output buffer too small"); goto done; }
    pos += vc2o_write_parse_info(out + pos, 0x00, (uint32_t)plen + 13, prev);
    memcpy(out + pos, payload, plen);
    pos += plen;
    prev = (uint32_t)plen + 13;
  }
  for (int frame = 0; frame < n_frames; ++frame)
   for (int pic = 0; pic < frame_pics; ++pic) {
    const uint8_t *src = raw + (size_t)frame * frame_bytes;
    /* Frame.cpp:90-104: first field = top field unless bottom field first */
    const int first_row = p->interlaced ? ((pic == 0) == !p->bottom_field_first ? 0 : 1) : 0;
    for (int c = 0; c < 3; ++c) {
      const int fh = c ? fch : p->height, w = c ? fcw : p->width;
      const int h = c ? ch : pp.height;
      const int ph = c ? g.chroma_h : g.luma_h, pw = c ? g.chroma_w : g.luma_w;
      vc2o_ingest(src, p->word_bytes, p->bit_depth, (size_t)fh * w, in);
      src += (size_t)fh * w * p->word_bytes;
      const int32_t *plane = in;
      if (p->interlaced) { take_field(in, fh, w, first_row, fld); plane = fld; }
      vc2o_pad(plane, h, w, tr[c], ph, pw);
      if ((rc = vc2o_dwt_forward(tr[c], ph, pw, p->kernel, p->depth))) goto done;
    }
    if (p->mode == 1) { /* HQ_CBR, EncodeStream.cpp:501-507 */
      vc2o_slice_bytes(g.y_slices, g.x_slices, picture_bytes, p->scalar, sbytes);
      if ((rc = vc2o_cbr_qindices(tr[0], tr[1], tr[2], &g, qm, sbytes, p->scalar, qidx))) goto done;
    } else if (p->mode == 0) {
      for (int i = 0; i < n_slices; ++i) qidx[i] = p->q_index;
    } else { /* LD, EncodeStream.cpp:511-517 */
      vc2o_slice_bytes(g.y_slices, g.x_slices, picture_bytes, 1, sbytes);
      if ((rc = vc2o_ld_qindices(tr[0], tr[1], tr[2], &g, qm, sbytes, qidx))) goto done;
    }
    for (int c = 0; c < 3; ++c) {
      const int ph = c ? g.chroma_h : g.luma_h, pw = c ? g.chroma_w : g.luma_w;
      rc = (p->mode == 2 ? vc2o_quantise_ld : vc2o_quantise_np)(tr[c], ph, pw, p->depth, qidx, g.y_slices, g.x_slices, qm, qc[c]);
      if (rc) goto done;
    }
    const uint32_t picture_number = (uint32_t)(pic + frame * frame_pics); /* Utils.cpp:52-63 */
    int num = picture_bytes, den = n_slices; /* utils::rationalise(pictureBytes, slices), EncodeStream.cpp:633 */
    { const int gg = gcd_i(num, den); if (gg) { num /= gg; den /= gg; } }
    /* UnsignedVLC holds its code word in 32 bits (VLC.h:27): values from 65535 up are outside the reference's domain */
    if (p->mode == 2 && (num >= 65535 || den >= 65535)) { rc = fail(VC2O_EINVAL, "oracle: LD slice bytes fraction does not fit a 32-bit exp-Golomb code"); goto done; }
    /* transform parameters, DataUnit.cpp:130-148 / :241-259 (whole picture: v3 flags by version;
     * fragments: always, :164-165 / :275-276) */
    uint8_t params[64];
    size_t params_len;
    {
      bitw w = {params, sizeof params, 0, 0, 0, 0, 0, 0};
      put_uvlc(&w, (uint32_t)p->kernel);
      put_uvlc(&w, (uint32_t)p->depth);
      if (fragmented || major >= 3) { put_bit(&w, 0); put_bit(&w, 0); }
      put_uvlc(&w, (uint32_t)g.x_slices);
      put_uvlc(&w, (uint32_t)g.y_slices);
      put_uvlc(&w, (uint32_t)(p->mode == 2 ? num : p->prefix));
      put_uvlc(&w, (uint32_t)(p->mode == 2 ? den : p->scalar));
      put_bit(&w, 0);
      w_align(&w);
      params_len = w.pos;
    }
    if (!fragmented) {
      /* picture data unit: DataUnit.cpp:236-266 (HQ) / :125-153 (LD) */
      if (pos + 13 + 4 + params_len > cap) { rc = fail(VC2O_ECAP, "output buffer too small"); goto done; }
      uint8_t *du = out + pos;
      put_be(du + 13, 4, picture_number);
      memcpy(du + 17, params, params_len);
      const size_t hlen = 4 + params_len;
      size_t dlen;
      if (p->mode == 2) rc = vc2o_ld_pack(qc[0], qc[1], qc[2], &g, qidx, sbytes, du + 13 + hlen, cap - pos - 13 - hlen, &dlen);
      else rc = vc2o_hq_pack(qc[0], qc[1], qc[2], &g, qidx, p->prefix, p->scalar, p->mode == 1 ? sbytes : NULL, du + 13 + hlen, cap - pos - 13 - hlen, &dlen);
      if (rc) goto done;
      const uint32_t next = (uint32_t)(hlen + dlen) + 13;
      vc2o_write_parse_info(du, p->mode == 2 ? 0xC8 : 0xE8, next, prev);
      prev = next;
      pos += next;
      continue;
    }
    /* fragments: DataUnit.cpp:156-232 (LD) / :267-342 (HQ) */
    const int code = p->mode == 2 ? 0xCC : 0xEC;
    const size_t need = (size_t)picture_bytes + (size_t)n_slices * ((size_t)p->prefix + p->scalar + 8) + 64;
    if (need > slices_cap) { slices = (uint8_t *)realloc(slices, need); slices_cap = need; }
    size_t dlen;
    if (p->mode == 2) rc = vc2o_ld_pack(qc[0], qc[1], qc[2], &g, qidx, sbytes, slices, slices_cap, &dlen);
    else rc = vc2o_hq_pack(qc[0], qc[1], qc[2], &g, qidx, p->prefix, p->scalar, sbytes, slices, slices_cap, &dlen);
    if (rc) goto done;
    if (pos + 13 + 8 + params_len > cap) { rc = fail(VC2O_ECAP, "output buffer too small"); goto done; }
    { /* first fragment: the transform parameters, slice count 0 */
      uint8_t *du = out + pos;
      const uint32_t next = (uint32_t)params_len + 8 + 13;
      vc2o_write_parse_info(du, code, next, prev);
      put_be(du + 13, 4, picture_number);
      put_be(du + 17, 2, (uint32_t)params_len);
      put_be(du + 19, 2, 0);
      memcpy(du + 21, params, params_len);
      prev = next;
      pos += next;
    }
    size_t spos = 0, frag_start = 0, frag_bytes = 0;
    int nslices = 0, ox = 0, oy = 0;
    for (int i = 0; i <= n_slices; ++i) {
      size_t ssize = 0;
      if (i < n_slices) {
        ssize = p->mode == 2 ? (size_t)sbytes[i] : hq_slice_size(slices + spos, dlen - spos, p->prefix, p->scalar);
        if (!ssize || spos + ssize > dlen) { rc = fail(VC2O_ESTREAM, "oracle: packed slices are inconsistent"); goto done; }
      }
      if (i == n_slices || (nslices > 0 && (long)(frag_bytes + ssize) > (long)p->fragment_length)) {
        if (pos + 13 + 12 + frag_bytes > cap) { rc = fail(VC2O_ECAP, "output buffer too small"); goto done; }
        uint8_t *du = out + pos;
        const uint32_t next = (uint32_t)frag_bytes + 12 + 13;
        vc2o_write_parse_info(du, code, next, prev);
        put_be(du + 13, 4, picture_number);
        put_be(du + 17, 2, (uint32_t)frag_bytes);
        put_be(du + 19, 2, (uint32_t)nslices);
        put_be(du + 21, 2, (uint32_t)ox);
        put_be(du + 23, 2, (uint32_t)oy);
        memcpy(du + 25, slices + frag_start, frag_bytes);
        prev = next;
        pos += next;
        ox = i % g.x_slices; oy = i / g.x_slices;
        nslices = 0; frag_start = spos; frag_bytes = 0;
      }
      frag_bytes += ssize;
      spos += ssize;
      ++nslices;
    }
  }
  if (pos + 13 > cap) { rc = fail(VC2O_ECAP, "output buffer too small"); goto done; }
  pos += vc2o_write_parse_info(out + pos, 0x10, 0, prev); /* DataUnit.cpp:366-370 */
  *out_len = pos;
done:
  free(in); free(fld); free(qidx); free(sbytes); free(slices);
  for (int c = 0; c < 3; ++c) { free(tr[c]); free(qc[c]); }
  return rc;
}

/* DecodeStream.cpp:103-992, -o Decoded: whole pictures and fragments, progressive and interlaced.
 * The sequence header is not interpreted (video parameters, including interlace and field order,
 * come from `p`); only its major version is read, for the v3 transform-parameter flags. */
int vc2o_decode_stream(const vc2o_params *p, const uint8_t *s, size_t len, uint8_t *raw_out,
                       size_t cap, int *n_frames_out) {
  vc2o_params pp = *p; /* the coded picture: a frame or one field (DecodeStream.cpp:318) */
  if (p->interlaced) pp.height = p->height / 2;
  vc2o_geom g;
  make_geom(&pp, &g, 1);
  int ch, cw, fch, fcw;
  vc2o_chroma_dims(pp.height, pp.width, pp.cf, &ch, &cw);
  vc2o_chroma_dims(p->height, p->width, p->cf, &fch, &fcw);
  const size_t pln = (size_t)g.luma_h * g.luma_w, pcn = (size_t)g.chroma_h * g.chroma_w;
  const size_t frame_bytes = ((size_t)p->height * p->width + 2 * (size_t)fch * fcw) * p->word_bytes;
  int32_t *q[3], *t[3];
  for (int c = 0; c < 3; ++c) {
    q[c] = (int32_t *)malloc(sizeof(int32_t) * (c ? pcn : pln));
    t[c] = (int32_t *)malloc(sizeof(int32_t) * (c ? pcn : pln));
  }
  int32_t *qidx = NULL, *sbytes = NULL, *crop = (int32_t *)malloc(sizeof(int32_t) * (size_t)pp.height * pp.width);
  uint8_t *field = (uint8_t *)malloc((size_t)pp.height * pp.width * p->word_bytes);
  /* fragment reassembly (one picture at a time is enough for streams in coding order) */
  uint32_t fr_picture = 0;
  int fr_open = 0, fr_ld = 0, fr_kernel = 0, fr_depth = 0, fr_a = 0, fr_b = 0, fr_decoded = 0, fr_slices = 0;
  const uint8_t **fr_ptr = NULL;
  size_t *fr_len = NULL;
  uint8_t *fr_payload = NULL;
  int rc = 0, frames = 0, major = 2, pic = 0;
  size_t pos = 0;
  while (pos + 13 <= len) {
    if (memcmp(s + pos, "BBCD", 4)) { rc = fail(VC2O_ESTREAM, "Read bytes do not match expected parse_info_header"); break; }
    const int code = s[pos + 4];
    const uint32_t next = ((uint32_t)s[pos + 5] << 24) | ((uint32_t)s[pos + 6] << 16) | ((uint32_t)s[pos + 7] << 8) | s[pos + 8];
    const uint8_t *body = s + pos + 13;
    const size_t body_len = (next ? next : 13) - 13;
    /* a complete picture's slice bytes, once found */
    const uint8_t *data = NULL;
    size_t data_len = 0;
    int ld = 0, kernel = 0, depth = 0, a = 0, b = 0;
    if (code == 0x00) {
      bitr r = {body, body_len, 0, 0, 0, 0, 0, 0};
      major = (int)get_uvlc(&r);
    } else if (code == 0xE8 || code == 0xC8) {
      bitr r = {body, len - pos - 13, 0, 0, 0, 0, 0, 0};
      get_bytes(&r, 4);
      kernel = (int)get_uvlc(&r); depth = (int)get_uvlc(&r);
      if (major >= 3) { get_bit(&r); get_bit(&r); }
      g.x_slices = (int)get_uvlc(&r);
      g.y_slices = (int)get_uvlc(&r);
      a = (int)get_uvlc(&r); b = (int)get_uvlc(&r); /* prefix,scalar | numer,denom */
      get_bit(&r);
      r_align(&r);
      ld = code == 0xC8;
      data = body + r.pos;
      data_len = len - pos - 13 - r.pos;
    } else if (code == 0xEC || code == 0xCC) { /* DecodeStream.cpp:614-797 / :799-977 */
      bitr r = {body, len - pos - 13, 0, 0, 0, 0, 0, 0};
      const uint32_t picnum = get_bytes(&r, 4);
      get_bytes(&r, 2);
      const int count = (int)get_bytes(&r, 2);
      if (count == 0) {
        fr_kernel = (int)get_uvlc(&r); fr_depth = (int)get_uvlc(&r);
        if (major >= 3) { get_bit(&r); get_bit(&r); }
        g.x_slices = (int)get_uvlc(&r);
        g.y_slices = (int)get_uvlc(&r);
        fr_a = (int)get_uvlc(&r); fr_b = (int)get_uvlc(&r);
        get_bit(&r);
        r_align(&r);
        fr_open = 1; fr_ld = code == 0xCC; fr_picture = picnum; fr_decoded = 0;
        fr_slices = g.x_slices * g.y_slices;
        fr_ptr = (const uint8_t **)realloc((void *)fr_ptr, sizeof(*fr_ptr) * fr_slices);
        fr_len = (size_t *)realloc(fr_len, sizeof(size_t) * fr_slices);
        memset(fr_len, 0, sizeof(size_t) * fr_slices);
        if (fr_ld) { /* DecodeStream.cpp:641, :664-665 */
          sbytes = (int32_t *)realloc(sbytes, sizeof(int32_t) * fr_slices);
          const int compressed = (fr_a * fr_slices) / fr_b;
          vc2o_slice_bytes(g.y_slices, g.x_slices, p->interlaced ? compressed / 2 : compressed, 1, sbytes);
        }
      } else if (fr_open && picnum == fr_picture) {
        const int ox = (int)get_bytes(&r, 2), oy = (int)get_bytes(&r, 2);
        const uint8_t *sp = body + r.pos;
        size_t left = body_len > r.pos ? body_len - r.pos : 0;
        int si = oy * g.x_slices + ox;
        for (int k = 0; k < count && si < fr_slices; ++k, ++si) { /* Slices.cpp:662-694 */
          const size_t sz = fr_ld ? (size_t)sbytes[si] : hq_slice_size(sp, left, fr_a, fr_b);
          if (!sz || sz > left) { rc = fail(VC2O_ESTREAM, "oracle: fragment does not hold its slices"); break; }
          fr_ptr[si] = sp; fr_len[si] = sz;
          sp += sz; left -= sz;
        }
        if (rc) break;
        fr_decoded += count;
        if (fr_decoded >= fr_slices) {
          size_t total = 0;
          for (int i = 0; i < fr_slices; ++i) total += fr_len[i];
          fr_payload = (uint8_t *)realloc(fr_payload, total + 1);
          total = 0;
          for (int i = 0; i < fr_slices; ++i) { memcpy(fr_payload + total, fr_ptr[i], fr_len[i]); total += fr_len[i]; }
          data = fr_payload; data_len = total;
          ld = fr_ld; kernel = fr_kernel; depth = fr_depth; a = fr_a; b = fr_b;
          fr_open = 0;
        }
      }
    }
    if (data) {
      if (depth != p->depth) { rc = fail(VC2O_ESTREAM, "oracle: stream depth differs from params"); break; }
      const int n_slices = g.x_slices * g.y_slices;
      qidx = (int32_t *)realloc(qidx, sizeof(int32_t) * n_slices);
      int32_t qm[3 * 31 + 1];
      if ((rc = vc2o_quant_matrix(kernel, depth, qm))) break;
      size_t used;
      if (!ld) {
        rc = vc2o_hq_unpack(data, data_len, &g, a, b, q[0], q[1], q[2], qidx, &used);
      } else { /* DecodeStream.cpp:312, :331-333: the reference halves the budget again for interlaced streams */
        sbytes = (int32_t *)realloc(sbytes, sizeof(int32_t) * n_slices);
        if (b == 0) { rc = fail(VC2O_ESTREAM, "oracle: slice bytes denominator is zero"); break; }
        const int compressed = (a * g.y_slices * g.x_slices) / b;
        vc2o_slice_bytes(g.y_slices, g.x_slices, p->interlaced ? compressed / 2 : compressed, 1, sbytes);
        rc = vc2o_ld_unpack(data, data_len, &g, sbytes, q[0], q[1], q[2], qidx, &used);
      }
      if (rc) break;
      if ((size_t)(frames + 1) * frame_bytes > cap) { rc = fail(VC2O_ECAP, "output buffer too small"); break; }
      uint8_t *dst = raw_out + (size_t)frames * frame_bytes;
      /* DecodeStream.cpp:417-428: first field, second field, then the frame is complete */
      const int first_row = p->interlaced ? ((pic == 0) == !p->bottom_field_first ? 0 : 1) : 0;
      for (int c = 0; c < 3 && !rc; ++c) {
        const int ph = c ? g.chroma_h : g.luma_h, pw = c ? g.chroma_w : g.luma_w;
        const int h = c ? ch : pp.height, w = c ? cw : pp.width;
        const int fh = c ? fch : p->height;
        rc = (ld ? vc2o_dequantise_ld : vc2o_dequantise_np)(q[c], ph, pw, depth, qidx, g.y_slices, g.x_slices, qm, t[c]);
        if (rc) break;
        if ((rc = vc2o_dwt_inverse(t[c], ph, pw, kernel, depth))) break;
        for (int y = 0; y < h; ++y) memcpy(crop + (size_t)y * w, t[c] + (size_t)y * pw, sizeof(int32_t) * w);
        const size_t row = (size_t)w * p->word_bytes;
        if (p->interlaced) {
          vc2o_clip_emit(crop, (size_t)h * w, p->word_bytes, p->bit_depth, field);
          for (int y = first_row, r = 0; y < fh && r < h; y += 2, ++r) memcpy(dst + (size_t)y * row, field + (size_t)r * row, row);
        } else {
          vc2o_clip_emit(crop, (size_t)h * w, p->word_bytes, p->bit_depth, dst);
        }
        dst += (size_t)fh * row;
      }
      if (rc) break;
      if (p->interlaced && pic == 0) pic = 1;
      else { pic = 0; ++frames; }
    }
    if (code == 0x10 || next == 0) break;
    pos += next;
  }
  *n_frames_out = frames;
  for (int c = 0; c < 3; ++c) { free(q[c]); free(t[c]); }
  free(qidx); free(sbytes); free(crop); free(field); free((void *)fr_ptr); free(fr_len); free(fr_payload);
  return rc;
}
