// Cross-lane primitives of the register-resident attention rows (one wave per row, a lane owns "its" columns of every slot) -- the
// backward's copy of the helpers csrc/tgat.hip defines for the forward (tgat.hip is left as it is: its committed counter profiles are
// keyed on its text).  Pure data movement plus sums in a FIXED order (xor 32, 16, 8, 4, 2, 1), no LDS crossbar (ds_bpermute) anywhere.
#pragma once
#include "common.h"

namespace tgmx {
namespace lanes {

// value of lane `src` (wave-uniform) in every lane: one v_readlane_b32 into an SGPR
__device__ __forceinline__ float bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ int bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// the value of lane (lane ^ O): quad permutes for 1 and 2, half-row mirror + quad reverse for 4 (i ^ 7 ^ 3), a row rotation by 8, the
// gfx950 row-pair / half-wave swaps for 16 and 32
template <int O>
__device__ __forceinline__ float lane_xor(float v) {
  static_assert(O == 1 || O == 2 || O == 4 || O == 8 || O == 16 || O == 32, "a power of two below the wave size");
  if constexpr (O == 1) return dpp<0xB1>(v);
  else if constexpr (O == 2) return dpp<0x4E>(v);
  else if constexpr (O == 4) return dpp<0x1B>(dpp<0x141>(v));
  else if constexpr (O == 8) return dpp<0x128>(v);
  else if constexpr (O == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((lane_id() & 16) ? r[0] : r[1]);
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((lane_id() & 32) ? r[0] : r[1]);
  }
}
// sum / max over the lanes that differ in the bits >= LO (lanes j, j +- LO, ...: the slots of one head when lane = slot * H + head)
template <int LO>
__device__ __forceinline__ float butterfly_sum(float v) {
  if constexpr (LO <= 1) v += lane_xor<1>(v);
  if constexpr (LO <= 2) v += lane_xor<2>(v);
  if constexpr (LO <= 4) v += lane_xor<4>(v);
  if constexpr (LO <= 8) v += lane_xor<8>(v);
  if constexpr (LO <= 16) v += lane_xor<16>(v);
  return v + lane_xor<32>(v);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, lane_xor<1>(v));
  v = fmaxf(v, lane_xor<2>(v));
  v = fmaxf(v, lane_xor<4>(v));
  v = fmaxf(v, lane_xor<8>(v));
  v = fmaxf(v, lane_xor<16>(v));
  return fmaxf(v, lane_xor<32>(v));
}

// One step (stride HALF) of the reduce-scatter of N entries over the 64 lanes.  HALF >= N: nothing left to split, both partners add
// and keep all N; below that the lower partner keeps the first HALF of its 2 HALF live entries, the upper one the second.
template <int HALF, int N>
__device__ __forceinline__ void reduce_step(float (&P)[N], int lane) {
  if constexpr (HALF >= N) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if constexpr (HALF == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(P[i]), __float_as_uint(P[i]), false, false);
        P[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      } else if constexpr (HALF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(P[i]), __float_as_uint(P[i]), false, false);
        P[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      } else {
        P[i] = P[i] + lane_xor<HALF>(P[i]);
      }
    }
  } else if constexpr (HALF == 32) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {  // the swap exchanges the upper half of one register with the lower half of the other: a' + b' IS keep + recv
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(P[i]), __float_as_uint(P[i + 32]), false, false);
      P[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
  } else if constexpr (HALF == 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(P[i]), __float_as_uint(P[i + 16]), false, false);
      P[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
  } else {
    const bool upper = (lane & HALF) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
      const float send = upper ? P[i] : P[i + HALF];
      const float recv = lane_xor<HALF>(send);
      const float keep = upper ? P[i + HALF] : P[i];
      P[i] = keep + recv;
    }
  }
}
// sum over the 64 lanes of every P[j], j < N (a power of two in [4, 64]); afterwards lane L holds entry L mod N in P[0]
template <int N>
__device__ __forceinline__ float reduce_scatter(float (&P)[N], int lane) {
  static_assert(N == 4 || N == 8 || N == 16 || N == 32 || N == 64, "N must be a power of two in [4, 64]");
  reduce_step<32, N>(P, lane);
  reduce_step<16, N>(P, lane);
  reduce_step<8, N>(P, lane);
  reduce_step<4, N>(P, lane);
  reduce_step<2, N>(P, lane);
  reduce_step<1, N>(P, lane);
  return P[0];
}

// cos and sin of one Time2Vec argument with the range-reduction path chosen by the CALLER (once per row, wave-uniform; sincos_t2v
// votes inside every evaluation): SMALL = the float reduction (|x| < kCosSmallLimit for every argument of the row), else double.
template <bool SMALL>
__device__ __forceinline__ void sincos_path(float x, float& sn_out, float& cs_out) {
  float rf;
  int q;
  if constexpr (SMALL) {
    float k = __builtin_rintf(x * 0.636619772367581343f);
    float r = __fmaf_rn(-k, 1.57079637050628662109375f, x);
    r = __fmaf_rn(-k, -4.37113900018624283e-8f, r);
    const float adj = r > 0.78539819f ? 1.f : (r < -0.78539819f ? -1.f : 0.f);
    k += adj;
    r = __fmaf_rn(-adj, 1.57079637050628662109375f, r);
    r = __fmaf_rn(-adj, -4.37113900018624283e-8f, r);
    r = __fmaf_rn(-k, -1.71512449e-15f, r);
    rf = r;
    q = (int)k & 3;
  } else {
    const double xd = (double)x;
    const double kd = __builtin_rint(xd * 0.63661977236758134308);
    double r = __builtin_fma(-kd, 1.57079632679489655800e+00, xd);
    r = __builtin_fma(-kd, 6.12323399573676603587e-17, r);
    rf = (float)r;
    q = (int)((long long)kd & 3);
  }
  const float r2 = rf * rf;
  float sp = -1.9515295891e-4f;
  sp = __fmaf_rn(sp, r2, 8.3321608736e-3f);
  sp = __fmaf_rn(sp, r2, -1.6666654611e-1f);
  const float sn = __fmaf_rn(sp * r2, rf, rf);
  float cp = 2.443315711809948e-5f;
  cp = __fmaf_rn(cp, r2, -1.388731625493765e-3f);
  cp = __fmaf_rn(cp, r2, 4.166664568298827e-2f);
  const float cs = __fmaf_rn(cp * r2, r2, __fmaf_rn(-0.5f, r2, 1.0f));
  const float cv = (q & 1) ? sn : cs;
  cs_out = (q == 1 || q == 2) ? -cv : cv;
  const float sv = (q & 1) ? cs : sn;
  sn_out = (q >= 2) ? -sv : sv;
}

}  // namespace lanes
}  // namespace tgmx
