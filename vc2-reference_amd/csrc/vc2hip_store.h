// Element access to the coefficient store and the level planes.
//
// Two element types (DESIGN.md "Data layout in HBM"):
//   int32_t : the plain store (fine-grained API, LD profile, geometries outside the fast kernels)
//   int16_t : the HQ batch path.  Every coefficient of the reference is an `int`; the values that occur in
//             practice fit 16 bits (10-bit DD97 depth 4 stays below 2^14), but nothing proves it for an
//             arbitrary picture or an arbitrary stream.  So a value outside [-32767, 32767] is stored as the
//             sentinel -32768 and its true value goes to the WIDE plane (int32, same element index), which is
//             otherwise never touched.  Readers test for the sentinel (one min over the values of a load) and
//             fetch the wide value for exactly those elements: bit-exact for every input, half the bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VC2_ST_SENTINEL (-32768)

template <class ST> struct St; // element access for store type ST

template <> struct St<int32_t> {
  static constexpr bool narrow = false;
  static __device__ __forceinline__ void load4(const int32_t *p, const int32_t *, int (&e)[4]) {
    const int4 v = *(const int4 *)p;
    e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
  }
  static __device__ __forceinline__ void load8(const int32_t *p, const int32_t *, int (&e)[8]) {
    const int4 a = *(const int4 *)p, b = *(const int4 *)(p + 4);
    e[0] = a.x; e[1] = a.y; e[2] = a.z; e[3] = a.w; e[4] = b.x; e[5] = b.y; e[6] = b.z; e[7] = b.w;
  }
  static __device__ __forceinline__ int load1(const int32_t *p, const int32_t *) { return *p; }
  static __device__ __forceinline__ void store4(int32_t *p, int32_t *, int a, int b, int c, int d) {
    *(int4 *)p = make_int4(a, b, c, d);
  }
  static __device__ __forceinline__ void store8(int32_t *p, int32_t *, const int (&e)[8]) {
    *(int4 *)p = make_int4(e[0], e[1], e[2], e[3]);
    *(int4 *)(p + 4) = make_int4(e[4], e[5], e[6], e[7]);
  }
  static __device__ __forceinline__ void store1(int32_t *p, int32_t *, int v) { *p = v; }
};

// two ints -> two 16-bit halves of one dword (low half first in memory)
__device__ __forceinline__ unsigned vc2_pack16(int lo, int hi) {
  return __builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u);
}
__device__ __forceinline__ int vc2_lo16(unsigned w) { return (int)(w << 16) >> 16; }
__device__ __forceinline__ int vc2_hi16(unsigned w) { return (int)w >> 16; }

template <> struct St<int16_t> {
  static constexpr bool narrow = true;
  // the wide plane shares the element index: wide = wide_base + (p - narrow_base); callers pass the matching pointer
  static __device__ __forceinline__ void load4(const int16_t *p, const int32_t *w, int (&e)[4]) {
    const uint2 v = *(const uint2 *)p;
    e[0] = vc2_lo16(v.x); e[1] = vc2_hi16(v.x); e[2] = vc2_lo16(v.y); e[3] = vc2_hi16(v.y);
    if (min(min(e[0], e[1]), min(e[2], e[3])) == VC2_ST_SENTINEL) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (e[k] == VC2_ST_SENTINEL) e[k] = w[k];
    }
  }
  static __device__ __forceinline__ void unpack8(const uint4 v, const int32_t *w, int (&e)[8]) {
    e[0] = vc2_lo16(v.x); e[1] = vc2_hi16(v.x); e[2] = vc2_lo16(v.y); e[3] = vc2_hi16(v.y);
    e[4] = vc2_lo16(v.z); e[5] = vc2_hi16(v.z); e[6] = vc2_lo16(v.w); e[7] = vc2_hi16(v.w);
    const int m = min(min(min(e[0], e[1]), min(e[2], e[3])), min(min(e[4], e[5]), min(e[6], e[7])));
    if (m == VC2_ST_SENTINEL) {
#pragma unroll
      for (int k = 0; k < 8; ++k) if (e[k] == VC2_ST_SENTINEL) e[k] = w[k];
    }
  }
  static __device__ __forceinline__ void load8(const int16_t *p, const int32_t *w, int (&e)[8]) {
    unpack8(*(const uint4 *)p, w, e);
  }
  static __device__ __forceinline__ int load1(const int16_t *p, const int32_t *w) {
    const int v = *p;
    return v == VC2_ST_SENTINEL ? *w : v;
  }
  static __device__ __forceinline__ bool fits(int v) { return (unsigned)(v + 32767) <= 65534u; }
  static __device__ __forceinline__ void store4(int16_t *p, int32_t *w, int a, int b, int c, int d) {
    const int mx = max(max(a, b), max(c, d)), mn = min(min(a, b), min(c, d));
    if (mx > 32767 || mn < -32767) {
      if (!fits(a)) { w[0] = a; a = VC2_ST_SENTINEL; }
      if (!fits(b)) { w[1] = b; b = VC2_ST_SENTINEL; }
      if (!fits(c)) { w[2] = c; c = VC2_ST_SENTINEL; }
      if (!fits(d)) { w[3] = d; d = VC2_ST_SENTINEL; }
    }
    *(uint2 *)p = make_uint2(vc2_pack16(a, b), vc2_pack16(c, d));
  }
  static __device__ __forceinline__ void store8(int16_t *p, int32_t *w, const int (&e0)[8]) {
    int e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = e0[k];
    const int mx = max(max(max(e[0], e[1]), max(e[2], e[3])), max(max(e[4], e[5]), max(e[6], e[7])));
    const int mn = min(min(min(e[0], e[1]), min(e[2], e[3])), min(min(e[4], e[5]), min(e[6], e[7])));
    if (mx > 32767 || mn < -32767) {
#pragma unroll
      for (int k = 0; k < 8; ++k) if (!fits(e[k])) { w[k] = e[k]; e[k] = VC2_ST_SENTINEL; }
    }
    *(uint4 *)p = make_uint4(vc2_pack16(e[0], e[1]), vc2_pack16(e[2], e[3]), vc2_pack16(e[4], e[5]), vc2_pack16(e[6], e[7]));
  }
  static __device__ __forceinline__ void store1(int16_t *p, int32_t *w, int v) {
    if (!fits(v)) { *w = v; v = VC2_ST_SENTINEL; }
    *p = (int16_t)v;
  }
};
