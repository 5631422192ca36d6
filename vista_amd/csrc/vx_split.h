// Exact three-way bf16 split of fp32 values (the arithmetic of spconv_w8s / spconv_cb16<NP> / wgrad_partial_s):
// x = h + m + l with h = x's top 16 bits, m = the top 16 bits of x - h, l = x - h - m (at most 8 significant bits:
// a bf16 without rounding). See the comment above spconv_w8s in vx_conv.hip.
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float vx_f32x4 __attribute__((ext_vector_type(4)));

struct Planes {
  bf16x8 h, m, l;
};
__device__ __forceinline__ unsigned top16(unsigned hi_word, unsigned lo_word) {  // {hi_word[31:16], lo_word[31:16]}
  return __builtin_amdgcn_perm(hi_word, lo_word, 0x07060302u);
}
// eight values, element i of every plane belongs to x[i]
__device__ __forceinline__ Planes split8s(const float (&x)[8]) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 ph, pm, pl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float xa = x[2 * i], xb = x[2 * i + 1];
    const unsigned ua = __float_as_uint(xa), ub = __float_as_uint(xb);
    ph[i] = top16(ub, ua);
    const float ra = xa - __uint_as_float(ua & 0xffff0000u), rb = xb - __uint_as_float(ub & 0xffff0000u);
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    pm[i] = top16(vb, va);
    const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
    pl[i] = top16(__float_as_uint(sb), __float_as_uint(sa));
  }
  Planes p;
  p.h = __builtin_bit_cast(bf16x8, ph);
  p.m = __builtin_bit_cast(bf16x8, pm);
  p.l = __builtin_bit_cast(bf16x8, pl);
  return p;
}
__device__ __forceinline__ Planes split8(const vx_f32x4& x0, const vx_f32x4& x1) {
  const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  return split8s(x);
}
