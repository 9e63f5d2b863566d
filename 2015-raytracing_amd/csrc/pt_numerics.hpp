// pt_numerics.hpp -- the numerics contract of the path on gfx950 (device side).
//
// OpenCL C leaves the bits of its built-in math to the implementation, so the "reference output" of A10 code.cl only exists
// once an implementation is named.  The one named (DESIGN.md "Numerics contract") is the one that exists for this hardware:
// AMD's own OpenCL C toolchain -- clang in OpenCL mode + the ROCm OpenCL built-in library (opencl.bc / ocml.bc), options
// -cl-std=CL1.2 -O3 -cl-fp32-correctly-rounded-divide-sqrt.  The reference's code.cl compiled that way (a gfx950 code object the
// test tree builds and runs on the MI355X itself, tests/test_ref_gpu.py) is what these kernels equal bit for bit:
//   + - *      IEEE binary32, RNE, denormals kept
//   a*b+c      fused EXACTLY where the OpenCL front end contracts (a product feeding a sum inside one expression; the left
//              product when both operands are products): written as explicit fmaf here, every TU builds with -ffp-contract=off
//   / , sqrt() correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt == -cl-fp32-correctly-rounded-divide-sqrt)
//   dot        fma(a.z,b.z, fma(a.y,b.y, a.x*b.x));  cross.x = fma(a.y,b.z, -(a.z*b.y)) ...     (opencl.bc)
//   normalize  v * v_rsq_f32(dot(v,v)) with the library's rescaling below 2^-126 / at inf; length = v_sqrt_f32(dot) likewise
//   min max fmin fmax   v_min_f32 / v_max_f32 (llvm.minnum / maxnum);  clamp = v_med3_f32;  mad = fma
//   sin / cos  ocml's __ocml_sin_f32 / __ocml_cos_f32 (restated below for |x| < 2^17, the library itself beyond)
// No fast-math anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef PT_EXACT_FAST_DIV
#define PT_EXACT_FAST_DIV 1   // 1: k_fusedPass<FAST> uses the refined-reciprocal forms below and defers the (rare) samples whose
                              // rays leave the guard window to k_fusedPass<EXACT>; 0: only the exact kernel runs
#endif
#ifndef PT_EXACT_FAST_NORM
#define PT_EXACT_FAST_NORM 1  // normalize(): per-lane guarded refined reciprocal (a zero-length vector must still give 1/0); 226.1 -> 223.7 ms
#endif

namespace pt {

#define PT_DEV __device__ __forceinline__

PT_DEV float cl_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_DEV float cl_min(float x, float y) { return __builtin_fminf(x, y); }    // v_min_f32: a NaN loses, -0 < +0
PT_DEV float cl_max(float x, float y) { return __builtin_fmaxf(x, y); }
PT_DEV float cl_fmin(float x, float y) { return __builtin_fminf(x, y); }
PT_DEV float cl_fmax(float x, float y) { return __builtin_fmaxf(x, y); }
PT_DEV float cl_clamp(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }   // v_med3_f32 (ockl median3)
// ocml rsqrt (normalize): v_rsq_f32, pre-scaled by 2^24 below 2^-126
PT_DEV float cl_rsqrt(float x) {
    const bool tiny = x < 0x1p-126f;
    const float r = __builtin_amdgcn_rsqf(tiny ? x * 16777216.0f : x);
    return tiny ? r * 4096.0f : r;
}
// llvm.sqrt !fpmath 3.0 as the gfx950 back end lowers it inside length(): v_sqrt_f32, pre-scaled by 2^32 below 2^-126
PT_DEV float cl_sqrt_approx(float x) {
    const bool tiny = x < 0x1p-126f;
    const float r = __builtin_amdgcn_sqrtf(tiny ? __builtin_ldexpf(x, 32) : x);
    return tiny ? __builtin_ldexpf(r, -16) : r;
}
PT_DEV float cl_fabs(float x) { return __builtin_fabsf(x); }
// Correctly rounded sqrt.  hipcc expands __builtin_sqrtf into: scale denormal inputs up (3 ops), v_sqrt_f32 (1 ulp), try the two
// neighbours s -+ 1 ulp against the residual (8 ops), scale back (2), patch 0 / inf by class (2).  The scaling exists because the
// residual fma(-s', s, x) underflows for small x; hipcc scales below 2^-96.  For every input with |x| outside (0, 2^-96) the 9-operation core
// alone returns the same bits -- 0, -0, inf, NaN and negatives included: the neighbour of 0 or inf is a NaN or a denormal whose
// residual test fails, so the v_sqrt result stands (checked on the device over all 2^32 bit patterns, mirt_debug_divcheck mode 5:
// the bare core differs from IEEE on 20.7 M patterns, every one of them with 0 < |x| < 2^-96 -- small positives, and the negative
// denormals, which v_sqrt_f32 flushes to -0 instead of answering NaN).  Those take the compiler's sequence.
#ifndef PT_EXACT_FAST_SQRT
#define PT_EXACT_FAST_SQRT 1
#endif
PT_DEV float sqrt_core(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float rdn = __builtin_fmaf(-dn, s, x), rup = __builtin_fmaf(-up, s, x);
    float r = (rdn <= 0.0f) ? dn : s;
    r = (rup > 0.0f) ? up : r;
    return r;
}
PT_DEV float cl_sqrt(float x) {
#if PT_PLAIN_DIV
    return __builtin_sqrtf(x);
#elif PT_EXACT_FAST_SQRT
    float r = sqrt_core(x);
    // 0 < |x| < 2^-96 (hipcc's own scaling threshold), as ONE unsigned compare on the bits of |x|: zero wraps to the top, NaN / inf sit above
    if (__builtin_expect((__float_as_uint(x) & 0x7FFFFFFFu) - 1u < 0x0F800000u - 1u, 0)) r = __builtin_sqrtf(x);   // rare lanes only
    return r;
#else
    return __builtin_sqrtf(x);
#endif
}
PT_DEV float cl_mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// float -> int: v_cvt_i32_f32 / v_cvt_u32_f32 themselves (truncate, saturate, NaN -> 0: the contract).  As inline assembly, because a C cast
// of an out-of-range value is poison to the compiler and the spelled-out version (three compares and selects around the cast) cost eight
// instructions where the hardware needs one.
PT_DEV int32_t f2i(float f) {
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}
PT_DEV uint32_t f2u(float f) {
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}
// The same conversion spelled in C, for WAVE-UNIFORM values (image size, lens grid side: kernel arguments): the result stays a scalar the
// compiler can branch and loop on without exec masks (an asm result is a per-lane value to it).
PT_DEV uint32_t f2u_uniform(float f) {
    if (f != f) return 0u;
    if (f >= 4294967296.0f) return UINT32_MAX;
    if (f <= 0.0f) return 0u;
    return (uint32_t)f;
}

// ---- exact division, cheaper ------------------------------------------------------------------------
// hipcc expands a correctly rounded x/y into v_div_scale x2, v_rcp, 2 fma (reciprocal refinement), mul, 4 fma
// (two quotient refinements, the last one v_div_fmas), v_div_fixup: 11 VALU ops, every time.  Two facts, both
// established ON THE DEVICE against that expansion (mirt_debug_divcheck / profiles/divcheck.py,
// tests/test_gpu_numerics.py), let the hot loops spend 3:
//  (1) rcp_refined(d) = fma(fma(-d, r0, 1), r0, r0), r0 = v_rcp_f32(d), equals 1.0f/d for EVERY float d except
//      +-0, +-inf, denormals and |d| >= 2^126 (checked over all 2^32 bit patterns; NaN gives NaN).
//  (2) with r = rcp_refined(d): q0 = n*r, q = fma(fma(-d, q0, n), r, q0) equals n/d for |d| in [2^-40, 2^40],
//      |n| in [2^-60, 2^60] (2^38 random pairs + every mantissa of d, zero mismatches); for n = +-0 the un-refined
//      product n*r already is the quotient (sign included) and is what div_exact3 returns.
// Outside those windows the affected lanes (and only they) redo the operation with the compiler's division.
#ifndef PT_PLAIN_DIV
#define PT_PLAIN_DIV 0   // 1 (with -fno-hip-fp32-correctly-rounded-divide-sqrt): every division and sqrt of the optimistic kernel as the compiler's
                         // 2.5-ulp forms -- the arithmetic of the reference built WITHOUT -cl-fp32-correctly-rounded-divide-sqrt.  A timing
                         // experiment only: no reference exists for its bits
#endif
// 1.0f / d as a value of its own.  Under the default contract (PT_PLAIN_DIV: 2.5 ulp allowed) the compiler is free to fold a reciprocal into the one product
// that uses it, and does so when there is only one -- the any-hit mesh kernel keeps t alone of the triangle test's three products and came out an ulp
// off the reference, whose 1 / div always has three uses (code.cl:260-277).  The empty asm keeps the quotient a quotient whatever its uses are.
PT_DEV float rcp_plain(float d) {
    float r = 1.0f / d;
#if PT_PLAIN_DIV
    asm volatile("" : "+v"(r));
#endif
    return r;
}
PT_DEV float rcp_refined(float d) {
#if PT_PLAIN_DIV
    return rcp_plain(d);
#endif
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
PT_DEV float div_shared(float n, float d, float r) {   // the compiler's five quotient operations, for the diagnostic
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}
PT_DEV float div_exact3(float n, float d, float r) {
#if PT_PLAIN_DIV
    return n / d;
#endif
    float q0 = n * r;
    float q = __builtin_fmaf(__builtin_fmaf(-d, q0, n), r, q0);
    return (n == 0.0f) ? q0 : q;
}
// div_exact3 for a quotient whose zero's SIGN nobody can see (it is converted to an integer, or only ever compared, added to and passed through
// min / max): without the select that hands a zero numerator's sign over -- for n = +-0 this returns a zero of either sign
PT_DEV float div_exact3_anyzero(float n, float d, float r) {
#if PT_PLAIN_DIV
    return n / d;
#endif
    const float q0 = n * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q0, n), r, q0);
}
PT_DEV bool rcp_window(float x) { return (__float_as_uint(x) & 0x7FFFFFFFu) - 0x00800000u < 0x7E800000u - 0x00800000u; }   // normal, < 2^126 (one unsigned compare)
// 1.0f/x, bit for bit.  `dont_care`: lanes whose result is never used (they must not force the slow path).
PT_DEV float rcp_exact(float x, bool dont_care = false) {
#if PT_EXACT_FAST_DIV && PT_EXACT_FAST_NORM
    float r = rcp_refined(x);
    if (__builtin_expect(!(dont_care || rcp_window(x)), 0)) r = rcp_plain(x);   // rare lanes only: s_cbranch_execz skips it
    return r;
#else
    return rcp_plain(x);
#endif
}

// sin and cos of one angle: ocml's __ocmlpriv_trigredsmall_f32 (gfx9+ branch: n = rint(|x| * 2/pi), three fused subtractions of
// n * pi/2) and __ocmlpriv_sincosred_f32 (fused minimax polynomials), one shared reduction.  |x| >= 2^17, inf and NaN take the
// library's own functions (Payne-Hanek reduction there): rare lanes only.
extern "C" __device__ float __ocml_sin_f32(float);
extern "C" __device__ float __ocml_cos_f32(float);
PT_DEV void cl_sincos(float x, float& sn, float& cs) {
    const float ax = __builtin_fabsf(x);
    const float n = __builtin_rintf(ax * 0x1.45f306p-1f);
    float r = __builtin_fmaf(n, -0x1.921fb4p+0f, ax);
    r = __builtin_fmaf(n, -0x1.4442dp-24f, r);
    r = __builtin_fmaf(n, -0x1.846988p-48f, r);
    const int q = (int)n & 3;
    const float s2 = r * r;
    float sp = __builtin_fmaf(s2, -0x1.983304p-13f, 0x1.110388p-7f);
    sp = __builtin_fmaf(s2, sp, -0x1.55553ap-3f);
    const float s = __builtin_fmaf(r, s2 * sp, r);
    float cp = __builtin_fmaf(s2, 0x1.aea668p-16f, -0x1.6c9e76p-10f);
    cp = __builtin_fmaf(s2, cp, 0x1.5557eep-5f);
    cp = __builtin_fmaf(s2, cp, -0x1.000008p-1f);
    const float c = __builtin_fmaf(s2, cp, 1.0f);
    const bool odd = (q & 1) != 0;
    const uint32_t flip = q > 1 ? 0x80000000u : 0u;
    sn = __uint_as_float(__float_as_uint(odd ? c : s) ^ flip ^ (__float_as_uint(x) & 0x80000000u));
    cs = __uint_as_float(__float_as_uint(odd ? -s : c) ^ flip);
    if (__builtin_expect(!(ax < 131072.0f), 0)) { sn = __ocml_sin_f32(x); cs = __ocml_cos_f32(x); }
}

}  // namespace pt
