// pt_numerics.hpp -- the numerics contract of the path on gfx950 (device side).
//
// OpenCL C leaves the bits of its built-in math to the implementation, so the
// "reference output" of A10 code.cl only exists once those are pinned.  The pin
// (DESIGN.md "Numerics contract") is:
//   + - *      IEEE binary32, RNE, never contracted  (-ffp-contract=off; every kernel TU)
//   / , sqrt   correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt:
//              v_div_scale/v_div_fmas/v_div_fixup and the refined v_sqrt sequence)
//   min / max  OpenCL common-function form  min(x,y) = y<x ? y : x,  max(x,y) = x<y ? y : x
//   fmin/fmax  IEEE minNum / maxNum (v_min_f32 / v_max_f32 in IEEE mode)
//   clamp      fmin(fmax(x,lo),hi)
//   mad        a*b + c, two roundings                     (A10 code.cl:209)
//   sin / cos  one shared Cody-Waite reduction + Cephes polynomials, fixed order
// Denormals are kept (gfx9+ default), no fast-math anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pt {

#define PT_DEV __device__ __forceinline__

PT_DEV float cl_min(float x, float y) { return (y < x) ? y : x; }
PT_DEV float cl_max(float x, float y) { return (x < y) ? y : x; }
PT_DEV float cl_fmin(float x, float y) { return __builtin_fminf(x, y); }
PT_DEV float cl_fmax(float x, float y) { return __builtin_fmaxf(x, y); }
PT_DEV float cl_clamp(float x, float lo, float hi) { return cl_fmin(cl_fmax(x, lo), hi); }
PT_DEV float cl_fabs(float x) { return __builtin_fabsf(x); }
PT_DEV float cl_sqrt(float x) { return __builtin_sqrtf(x); }
PT_DEV float cl_mad(float a, float b, float c) { return a * b + c; }

// float -> int the way v_cvt_i32_f32 does it (truncate, saturate, NaN -> 0), spelled
// out so the compiler cannot treat an out-of-range input as poison.
PT_DEV int32_t f2i(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
PT_DEV uint32_t f2u(float f) {
    if (f != f) return 0u;
    if (f >= 4294967296.0f) return UINT32_MAX;
    if (f <= 0.0f) return 0u;
    return (uint32_t)f;
}

// sin and cos of one angle.  k = rint(x*2/pi) by the 1.5*2^23 trick; r = x - k*pi/2 in
// three exact-product steps; Cephes sinf/cosf minimax polynomials on [-pi/4, pi/4].
// <= 1.5 ulp on [-pi/4, 3pi/4], the only range concentric_distort produces.
PT_DEV void cl_sincos(float x, float& sn, float& cs) {
    const float two_over_pi = 0.63661977236758134308f;
    const float magic = 12582912.0f;
    const float pio2_hi = 1.5703125f;
    const float pio2_md = 4.837512969970703125e-4f;
    const float pio2_lo = 7.54978995489188216e-8f;

    float kf = x * two_over_pi + magic;
    kf = kf - magic;
    int32_t q = (kf == kf) ? (int32_t)kf : 0;

    float r = x - kf * pio2_hi;
    r = r - kf * pio2_md;
    r = r - kf * pio2_lo;
    float r2 = r * r;

    float sp = -1.9515295891e-4f * r2;
    sp = sp + 8.3321608736e-3f;
    sp = sp * r2;
    sp = sp - 1.6666654611e-1f;
    sp = sp * r2;
    sp = sp * r;
    float s = sp + r;

    float cp = 2.443315711809948e-5f * r2;
    cp = cp - 1.388731625493765e-3f;
    cp = cp * r2;
    cp = cp + 4.166664568298827e-2f;
    cp = cp * r2;
    cp = cp * r2;
    float c = cp - 0.5f * r2;
    c = c + 1.0f;

    const bool swap = (q & 1) != 0;
    float a = swap ? c : s;   // |sin|
    float b = swap ? s : c;   // |cos|
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

}  // namespace pt
