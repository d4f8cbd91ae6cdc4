// x3_common.hpp -- f32 products on the bf16 matrix cores: exact three-way operand split and the six-product accumulate (kernels_x3.hip has the derivation).
#pragma once
#include <hip/hip_runtime.h>
#include "mfma_common.hpp"

namespace tnqs {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
// eight k-slots of one real MFMA operand as three bf16 pieces; slots (2 i, 2 i + 1) share register i
struct P3 { u4 h, m, l; };

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                                   // (upper half of x1, upper half of x0)
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u); // <= 8 significant bits left: exact
}
template <int MODE = 0>
__device__ __forceinline__ P3 split8(const float (&x)[8]) {
    P3 p;
    if (MODE == 1 || MODE == 3 || MODE == 4) {        // timing experiment: no splitting
#pragma unroll
        for (int i = 0; i < 4; ++i) { p.h[i] = __float_as_uint(x[2 * i]); p.m[i] = __float_as_uint(x[2 * i + 1]); p.l[i] = __float_as_uint(x[i]); }
        return p;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned h, m, l; split_pair(x[2 * i], x[2 * i + 1], h, m, l); p.h[i] = h; p.m[i] = m; p.l[i] = l; }
    return p;
}
__device__ __forceinline__ P3 neg(const P3& a) {
    P3 p;
#pragma unroll
    for (int i = 0; i < 4; ++i) { p.h[i] = a.h[i] ^ 0x80008000u; p.m[i] = a.m[i] ^ 0x80008000u; p.l[i] = a.l[i] ^ 0x80008000u; }
    return p;
}
#define TNQS_BF(A, B, ACC) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, A), __builtin_bit_cast(bf8, B), ACC, 0, 0, 0)
// two independent accumulators side by side (no instruction depends on the one in front of it):  p += a b,  q += c d,  six products each, small terms first
template <bool FIRST, int MODE = 0>
__device__ __forceinline__ void mac6x2(v16f& p, const P3& a, const P3& b, v16f& q, const P3& c, const P3& d) {
    const v16f z = (v16f)(0.f);
    if (MODE == 2) {        // timing experiment: no matrix instructions, the pieces stay alive
        if (FIRST) { p = z; q = z; }
        p[0] += __uint_as_float(a.h[0] ^ a.m[1] ^ a.l[2] ^ b.h[3] ^ b.m[0] ^ b.l[1]);
        q[0] += __uint_as_float(c.h[0] ^ c.m[1] ^ c.l[2] ^ d.h[3] ^ d.m[0] ^ d.l[1]);
        return;
    }
    p = TNQS_BF(a.l, b.h, FIRST ? z : p); q = TNQS_BF(c.l, d.h, FIRST ? z : q);
    p = TNQS_BF(a.h, b.l, p);             q = TNQS_BF(c.h, d.l, q);
    p = TNQS_BF(a.m, b.m, p);             q = TNQS_BF(c.m, d.m, q);
    p = TNQS_BF(a.m, b.h, p);             q = TNQS_BF(c.m, d.h, q);
    p = TNQS_BF(a.h, b.m, p);             q = TNQS_BF(c.h, d.m, q);
    p = TNQS_BF(a.h, b.h, p);             q = TNQS_BF(c.h, d.h, q);
}


}  // namespace tnqs
