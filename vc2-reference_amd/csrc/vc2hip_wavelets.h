// Wavelet definitions shared by the DWT kernels (generic and register-blocked).
#pragma once
#include "vc2hip_internal.h"

// ------------------------------------------------------------------------------------------
// wavelet definitions: nsteps, accuracy bits, halo (input samples per side) and the lifting
// deltas.  at(d) reads the opposite-parity sample at PAIR offset d from the target's pair index:
//   odd  target 2K+1, tap offset o (odd)  -> even sample pair K + (o+1)/2
//   even target 2K,   tap offset o (odd)  -> odd  sample pair K + (o-1)/2
// delta<K,S>() is what the FORWARD transform adds to the target; the inverse subtracts it.
// ------------------------------------------------------------------------------------------
template <int K> struct WT;
template <> struct WT<VC2HIP_DD97>     { static constexpr int nsteps = 2, accuracy = 1, halo = 4; };
template <> struct WT<VC2HIP_LEGALL>   { static constexpr int nsteps = 2, accuracy = 1, halo = 2; };
template <> struct WT<VC2HIP_DD137>    { static constexpr int nsteps = 2, accuracy = 1, halo = 6; };
template <> struct WT<VC2HIP_HAAR0>    { static constexpr int nsteps = 2, accuracy = 0, halo = 0; };
template <> struct WT<VC2HIP_HAAR1>    { static constexpr int nsteps = 2, accuracy = 1, halo = 0; };
template <> struct WT<VC2HIP_FIDELITY> { static constexpr int nsteps = 2, accuracy = 0, halo = 14; };
template <> struct WT<VC2HIP_DAUB97>   { static constexpr int nsteps = 4, accuracy = 1, halo = 4; };

template <int K, int S> __device__ __forceinline__ constexpr bool step_targets_odd() {
  if constexpr (K == VC2HIP_FIDELITY) return S == 1;        // update (even) first, then predict
  else return (S % 2) == 0;                                  // predict (odd) first
}

// 9 * x as shift-and-add (one v_lshl_add_u32; the integer multiply is a quarter-rate instruction)
__device__ __forceinline__ int vc2_times9(int x) { return (int)(((unsigned)x << 3) + (unsigned)x); }
template <class T> __device__ __forceinline__ T vc2_times9(const T &x) { const T d = x + x, q = d + d; return q + q + x; }

template <int K, int S, class F> __device__ __forceinline__ auto lift_delta(F at) {
  if constexpr (K == VC2HIP_DD97 || K == VC2HIP_DD137) {
    if constexpr (S == 0) return -((vc2_times9(at(0) + at(1)) - (at(-1) + at(2)) + 8) >> 4);
    else if constexpr (K == VC2HIP_DD97) return (at(-1) + at(0) + 2) >> 2;
    else return (vc2_times9(at(-1) + at(0)) - (at(-2) + at(1)) + 16) >> 5;
  } else if constexpr (K == VC2HIP_LEGALL) {
    if constexpr (S == 0) return -((at(0) + at(1) + 1) >> 1);
    else return (at(-1) + at(0) + 2) >> 2;
  } else if constexpr (K == VC2HIP_HAAR0 || K == VC2HIP_HAAR1) {
    if constexpr (S == 0) return -at(0);
    else return (at(0) + 1) >> 1;
  } else if constexpr (K == VC2HIP_FIDELITY) {
    if constexpr (S == 0)
      return (-8 * at(-4) + 21 * at(-3) - 46 * at(-2) + 161 * at(-1) + 161 * at(0) - 46 * at(1) +
              21 * at(2) - 8 * at(3) + 128) >> 8;
    else
      return -((-2 * at(-3) + 10 * at(-2) - 25 * at(-1) + 81 * at(0) + 81 * at(1) - 25 * at(2) +
                10 * at(3) - 2 * at(4) + 128) >> 8);
  } else { // Daub97
    if constexpr (S == 0) return -((6497 * at(0) + 6497 * at(1) + 2048) >> 12);
    else if constexpr (S == 1) return -((217 * at(-1) + 217 * at(0) + 2048) >> 12);
    else if constexpr (S == 2) return (3616 * at(0) + 3616 * at(1) + 2048) >> 12;
    else return (1817 * at(-1) + 1817 * at(0) + 2048) >> 12;
  }
}

template <int K> __host__ __device__ constexpr int halo_y() { return WT<K>::halo; }
template <int K> __host__ __device__ constexpr int halo_x() { return (WT<K>::halo + 7) & ~7; }


// pair-offset reach [dmin, dmax] of the taps of step S (see lift_delta)
template <int K, int S> __host__ __device__ constexpr int step_dmin() {
  if constexpr (K == VC2HIP_DD97) return S == 0 ? -1 : -1;
  else if constexpr (K == VC2HIP_LEGALL) return S == 0 ? 0 : -1;
  else if constexpr (K == VC2HIP_DD137) return S == 0 ? -1 : -2;
  else if constexpr (K == VC2HIP_HAAR0 || K == VC2HIP_HAAR1) return 0;
  else if constexpr (K == VC2HIP_FIDELITY) return S == 0 ? -4 : -3;
  else return (S % 2 == 0) ? 0 : -1;
}
template <int K, int S> __host__ __device__ constexpr int step_dmax() {
  if constexpr (K == VC2HIP_DD97) return S == 0 ? 2 : 0;
  else if constexpr (K == VC2HIP_LEGALL) return S == 0 ? 1 : 0;
  else if constexpr (K == VC2HIP_DD137) return S == 0 ? 2 : 1;
  else if constexpr (K == VC2HIP_HAAR0 || K == VC2HIP_HAAR1) return 0;
  else if constexpr (K == VC2HIP_FIDELITY) return S == 0 ? 3 : 4;
  else return (S % 2 == 0) ? 1 : 0;
}
