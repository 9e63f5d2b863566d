// oracle/ref/cl_runtime_shim.cpp -- TEST INFRASTRUCTURE, build-container only.
//
// Lets the reference's OpenCL C kernels (compiled for x86 by `clang -x cl`
// straight from /root/reference, see ../Makefile) run as a sequential CPU
// program.  It supplies exactly what an OpenCL implementation would supply at
// run time and the image does not have for x86:
//
//   1. get_global_id()                      -- the NDRange index
//   2. the 18 OpenCL built-in math functions the kernels call (nm -u of the
//      compiled object): dot cross normalize length distance fabs(float3)
//      sqrt sin cos fabs fmin fmax min max mad clamp clamp(float4,...)
//   3. a work-item loop per kernel (ref_<assign>_<kernel>), row-major for the
//      2-D initTrace launch -- that ORDER is the oracle's definition of the
//      reference's seeds[col] race (A10 code.cl:429 vs :469-470).
//
// The kernel bodies are the reference's, unmodified, compiled with the OpenCL
// front end's default contraction (-ffp-contract=on, -mfma: every llvm.fmuladd
// becomes one fused operation, as on gfx950).  The built-ins are the CPU MODEL
// (../cl_numerics.h) of AMD's OpenCL C library on gfx950.  This x86 build is the
// container-side twin of the real pin, oracle/_ref/a10_gfx950.hsaco (the same
// text compiled by AMD's OpenCL toolchain and run on the MI355X, oracle/ref_gpu.py):
// tests/test_ref_gpu.py holds the two equal, bit for bit, on every fixture.
//
// Nothing in here is shipped or timed; the .so lands in oracle/_ref/.

#include <cstdint>
#include <cstring>
#include "../cl_numerics.h"

typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef float float16 __attribute__((ext_vector_type(16)));
typedef unsigned char uchar4 __attribute__((ext_vector_type(4)));

// ---- 1. NDRange index ------------------------------------------------------
// Every definition below carries the Itanium-mangled name the OpenCL front end
// emitted for the call (nm -u of the compiled object) as an explicit asm
// label, so nothing here collides with <math.h>.
#define CL_SYM(s) __asm__(s)
static thread_local size_t g_gid[3];
size_t cl_get_global_id(unsigned int d) CL_SYM("_Z13get_global_idj");
size_t cl_get_global_id(unsigned int d) { return d < 3 ? g_gid[d] : 0; }

// ---- 2. built-ins -----------------------------------------------------------
float cl_sqrt(float) CL_SYM("_Z4sqrtf");
float cl_sin(float) CL_SYM("_Z3sinf");
float cl_cos(float) CL_SYM("_Z3cosf");
float cl_fabs(float) CL_SYM("_Z4fabsf");
float cl_fmin(float, float) CL_SYM("_Z4fminff");
float cl_fmax(float, float) CL_SYM("_Z4fmaxff");
float cl_min(float, float) CL_SYM("_Z3minff");
float cl_max(float, float) CL_SYM("_Z3maxff");
float cl_mad(float, float, float) CL_SYM("_Z3madfff");
float cl_clamp(float, float, float) CL_SYM("_Z5clampfff");
float4 cl_clamp4(float4, float, float) CL_SYM("_Z5clampDv4_fff");
float3 cl_fabs3(float3) CL_SYM("_Z4fabsDv3_f");
float cl_dot(float3, float3) CL_SYM("_Z3dotDv3_fS_");
float3 cl_cross(float3, float3) CL_SYM("_Z5crossDv3_fS_");
float cl_length(float3) CL_SYM("_Z6lengthDv3_f");
float cl_distance(float3, float3) CL_SYM("_Z8distanceDv3_fS_");
float3 cl_normalize(float3) CL_SYM("_Z9normalizeDv3_f");

float cl_sqrt(float x) { return cln_sqrt(x); }
float cl_sin(float x) { return cln_sin(x); }
float cl_cos(float x) { return cln_cos(x); }
float cl_fabs(float x) { return cln_fabs(x); }
float cl_fmin(float a, float b) { return cln_fmin(a, b); }
float cl_fmax(float a, float b) { return cln_fmax(a, b); }
float cl_min(float a, float b) { return cln_min(a, b); }
float cl_max(float a, float b) { return cln_max(a, b); }
float cl_mad(float a, float b, float c) { return cln_mad(a, b, c); }
float cl_clamp(float x, float lo, float hi) { return cln_clamp(x, lo, hi); }
float4 cl_clamp4(float4 v, float lo, float hi) {
    float4 r;
    r.x = cln_clamp(v.x, lo, hi);
    r.y = cln_clamp(v.y, lo, hi);
    r.z = cln_clamp(v.z, lo, hi);
    r.w = cln_clamp(v.w, lo, hi);
    return r;
}
float3 cl_fabs3(float3 v) {
    float3 r;
    r.x = cln_fabs(v.x);
    r.y = cln_fabs(v.y);
    r.z = cln_fabs(v.z);
    return r;
}
float cl_dot(float3 a, float3 b) { return cln_dot3(a.x, a.y, a.z, b.x, b.y, b.z); }
float3 cl_cross(float3 a, float3 b) {
    const float A[3] = {a.x, a.y, a.z}, B[3] = {b.x, b.y, b.z};
    float r[3];
    cln_cross3(A, B, r);
    return (float3){r[0], r[1], r[2]};
}
float cl_length(float3 a) { return cln_length3(a.x, a.y, a.z); }
float cl_distance(float3 a, float3 b) { return cln_length3(a.x - b.x, a.y - b.y, a.z - b.z); }
float3 cl_normalize(float3 a) {
    const float A[3] = {a.x, a.y, a.z};
    float r[3];
    cln_normalize3(A, r);
    return (float3){r[0], r[1], r[2]};
}
extern "C" void ref_set_hw_tables(const signed char* rsq_delta, const signed char* sqrt_delta) { cln_hw_tables((const int8_t*)rsq_delta, (const int8_t*)sqrt_delta); }

// ---- 3. kernels of the compiled reference + work-item loops ---------------
struct AABB { float3 pmin; float3 pmax; };
struct Ray;
struct Poi;

static inline AABB mkbox(const float* b8) {
    AABB a;
    a.pmin = (float3){b8[0], b8[1], b8[2]};
    a.pmax = (float3){b8[4], b8[5], b8[6]};
    return a;
}
static inline float16 mk16(const float* f) {
    float16 v;
    for (int i = 0; i < 16; ++i) v[i] = f[i];
    return v;
}

#define FOR_1D(n) for (size_t _i = 0; _i < (size_t)(n); ++_i) if ((g_gid[0] = _i, g_gid[1] = 0, g_gid[2] = 0, true))

#ifdef REF_A10
extern "C" {
void __clang_ocl_kern_imp_sizeofRay(unsigned*);
void __clang_ocl_kern_imp_sizeofPoi(unsigned*);
void __clang_ocl_kern_imp_initAcu(float4*, unsigned);
void __clang_ocl_kern_imp_initTrace(int*, Ray*, Poi*, AABB, float16, float, float, unsigned);
void __clang_ocl_kern_imp_bouncePaths(Poi*, Ray*, int*, unsigned);
void __clang_ocl_kern_imp_lightRender(Poi*, Ray*, float4*, float16, unsigned);
void __clang_ocl_kern_imp_initShadowTrace(Ray*, Poi*, unsigned, float16, int*);
void __clang_ocl_kern_imp_sphereTrace(unsigned, Poi*, Ray*, float4*, unsigned*, unsigned*, AABB, unsigned);
void __clang_ocl_kern_imp_triangleTrace(unsigned, Poi*, Ray*, float3*, float3*, unsigned*, unsigned*, AABB, unsigned);
void __clang_ocl_kern_imp_meshTrace(unsigned, Poi*, Ray*, float3*, float3*, unsigned*, unsigned, AABB, unsigned);
void __clang_ocl_kern_imp_sphereShadowTrace(unsigned, Ray*, float4*, unsigned*, AABB, unsigned);
void __clang_ocl_kern_imp_triangleShadowTrace(unsigned, Ray*, float3*, unsigned*, AABB, unsigned);
void __clang_ocl_kern_imp_sceneRender(float4*, Poi*, Ray*, float4*, float16, unsigned);
void __clang_ocl_kern_imp_copyToPixel(uchar4*, float4*, float, unsigned, unsigned);
int ref_rand(int*);

// Exported entry points: plain pointers, float16 as const float[16], AABB as
// const float[8] in the host's (min,1,max,1) packing (A10 code.js:610-621).
// `gsz` is the (padded) global size the host would enqueue.
unsigned ref_a10_sizeofRay(void) { unsigned s = 0; g_gid[0] = 0; __clang_ocl_kern_imp_sizeofRay(&s); return s; }
unsigned ref_a10_sizeofPoi(void) { unsigned s = 0; g_gid[0] = 0; __clang_ocl_kern_imp_sizeofPoi(&s); return s; }
int ref_a10_rand(int* seed) { return ref_rand(seed); }

void ref_a10_initAcu(void* acu, unsigned total, size_t gsz) {
    FOR_1D(gsz) __clang_ocl_kern_imp_initAcu((float4*)acu, total);
}
void ref_a10_initTrace(int* seeds, void* rays, void* pois, const float* bound, const float* cam,
                       float focal, float lens_rad, unsigned rpp, size_t gx, size_t gy) {
    AABB b = mkbox(bound);
    float16 c = mk16(cam);
    for (size_t row = 0; row < gy; ++row)
        for (size_t col = 0; col < gx; ++col) {
            g_gid[0] = col; g_gid[1] = row; g_gid[2] = 0;
            __clang_ocl_kern_imp_initTrace(seeds, (Ray*)rays, (Poi*)pois, b, c, focal, lens_rad, rpp);
        }
}
void ref_a10_bouncePaths(void* pois, void* rays, int* seeds, unsigned total, size_t gsz) {
    FOR_1D(gsz) __clang_ocl_kern_imp_bouncePaths((Poi*)pois, (Ray*)rays, seeds, total);
}
void ref_a10_lightRender(void* pois, void* rays, void* acu, const float* light, unsigned total, size_t gsz) {
    float16 l = mk16(light);
    FOR_1D(gsz) __clang_ocl_kern_imp_lightRender((Poi*)pois, (Ray*)rays, (float4*)acu, l, total);
}
void ref_a10_initShadowTrace(void* shadow, void* pois, unsigned total, const float* light, int* seeds, size_t gsz) {
    float16 l = mk16(light);
    FOR_1D(gsz) __clang_ocl_kern_imp_initShadowTrace((Ray*)shadow, (Poi*)pois, total, l, seeds);
}
void ref_a10_sphereTrace(unsigned total, void* pois, void* rays, void* spheres, unsigned* matid,
                         unsigned* box, const float* bound, unsigned n, size_t gsz) {
    AABB b = mkbox(bound);
    FOR_1D(gsz) __clang_ocl_kern_imp_sphereTrace(total, (Poi*)pois, (Ray*)rays, (float4*)spheres, matid, box, b, n);
}
void ref_a10_triangleTrace(unsigned total, void* pois, void* rays, void* pos, void* nor, unsigned* matid,
                           unsigned* box, const float* bound, unsigned n, size_t gsz) {
    AABB b = mkbox(bound);
    FOR_1D(gsz) __clang_ocl_kern_imp_triangleTrace(total, (Poi*)pois, (Ray*)rays, (float3*)pos, (float3*)nor, matid, box, b, n);
}
void ref_a10_meshTrace(unsigned total, void* pois, void* rays, void* pos, void* nor, unsigned* box,
                       unsigned matid, const float* bound, unsigned n, size_t gsz) {
    AABB b = mkbox(bound);
    FOR_1D(gsz) __clang_ocl_kern_imp_meshTrace(total, (Poi*)pois, (Ray*)rays, (float3*)pos, (float3*)nor, box, matid, b, n);
}
void ref_a10_sphereShadowTrace(unsigned total, void* shadow, void* spheres, unsigned* box,
                               const float* bound, unsigned n, size_t gsz) {
    AABB b = mkbox(bound);
    FOR_1D(gsz) __clang_ocl_kern_imp_sphereShadowTrace(total, (Ray*)shadow, (float4*)spheres, box, b, n);
}
void ref_a10_triangleShadowTrace(unsigned total, void* shadow, void* pos, unsigned* box,
                                 const float* bound, unsigned n, size_t gsz) {
    AABB b = mkbox(bound);
    FOR_1D(gsz) __clang_ocl_kern_imp_triangleShadowTrace(total, (Ray*)shadow, (float3*)pos, box, b, n);
}
void ref_a10_sceneRender(void* acu, void* pois, void* shadow, void* material, const float* light,
                         unsigned total, size_t gsz) {
    float16 l = mk16(light);
    FOR_1D(gsz) __clang_ocl_kern_imp_sceneRender((float4*)acu, (Poi*)pois, (Ray*)shadow, (float4*)material, l, total);
}
void ref_a10_copyToPixel(void* pixel, void* acu, float m, unsigned pixels, unsigned rpp, size_t gsz) {
    FOR_1D(gsz) __clang_ocl_kern_imp_copyToPixel((uchar4*)pixel, (float4*)acu, m, pixels, rpp);
}

// built-in probes, so the plain-C restatement can be checked against the very
// functions the compiled kernels called
float ref_bi_sin(float x) { return cl_sin(x); }
float ref_bi_cos(float x) { return cl_cos(x); }
}  // extern "C"
#endif  // REF_A10

// ---- single-frame kernels of the earlier assignments: 2-D NDRange, row-major work-item order ----
#define FOR_2D(gx, gy) for (size_t _r = 0; _r < (size_t)(gy); ++_r) for (size_t _c = 0; _c < (size_t)(gx); ++_c) \
    if ((g_gid[0] = _c, g_gid[1] = _r, g_gid[2] = 0, true))

#ifdef REF_A01
extern "C" {
void __clang_ocl_kern_imp_raytrace(uchar4*, float16);
// the reference kernel has no range check: launch it on exactly cols x rows (the reference pads the NDRange and
// relies on the canvas being a multiple of the work-group shape, A01 code.js:237-241)
void ref_a01_raytrace(void* pixels, const float* cam, size_t gx, size_t gy) {
    float16 c = mk16(cam);
    FOR_2D(gx, gy) __clang_ocl_kern_imp_raytrace((uchar4*)pixels, c);
}
}
#endif

#ifdef REF_A04
extern "C" {
void __clang_ocl_kern_imp_sizeofRay(unsigned*);
void __clang_ocl_kern_imp_initTrace(uchar4*, float16, Ray*);
void __clang_ocl_kern_imp_meshTrace(uchar4*, float16, Ray*, unsigned, float3*, float3*, unsigned*, float4*);
unsigned ref_a04_sizeofRay(void) { unsigned s = 0; g_gid[0] = 0; __clang_ocl_kern_imp_sizeofRay(&s); return s; }
void ref_a04_initTrace(void* pixels, const float* cam, void* rays, size_t gx, size_t gy) {
    float16 c = mk16(cam);
    FOR_2D(gx, gy) __clang_ocl_kern_imp_initTrace((uchar4*)pixels, c, (Ray*)rays);
}
void ref_a04_meshTrace(void* pixels, const float* cam, void* rays, unsigned t_size, void* pos, void* nor, unsigned* mindex,
                       void* mcolor, size_t gx, size_t gy) {
    float16 c = mk16(cam);
    FOR_2D(gx, gy) __clang_ocl_kern_imp_meshTrace((uchar4*)pixels, c, (Ray*)rays, t_size, (float3*)pos, (float3*)nor, mindex, (float4*)mcolor);
}
}
#endif

#ifdef REF_A07
extern "C" {
void __clang_ocl_kern_imp_sizeofRay(unsigned*);
void __clang_ocl_kern_imp_initTrace(uchar4*, float16, Ray*, AABB);
void __clang_ocl_kern_imp_meshTrace(uchar4*, float16, Ray*, unsigned, float3*, float3*, unsigned*, float4*, AABB, unsigned, unsigned*);
void __clang_ocl_kern_imp_molTrace(uchar4*, float16, Ray*, unsigned, float4*, unsigned*, float4*, AABB, unsigned, unsigned*);
unsigned ref_a07_sizeofRay(void) { unsigned s = 0; g_gid[0] = 0; __clang_ocl_kern_imp_sizeofRay(&s); return s; }
void ref_a07_initTrace(void* pixels, const float* cam, void* rays, const float* bound, size_t gx, size_t gy) {
    float16 c = mk16(cam);
    AABB b = mkbox(bound);
    FOR_2D(gx, gy) __clang_ocl_kern_imp_initTrace((uchar4*)pixels, c, (Ray*)rays, b);
}
void ref_a07_meshTrace(void* pixels, const float* cam, void* rays, unsigned t_size, void* pos, void* nor, unsigned* mindex,
                       void* mcolor, const float* bound, unsigned n_slabs, unsigned* slab_size, size_t gx, size_t gy) {
    float16 c = mk16(cam);
    AABB b = mkbox(bound);
    FOR_2D(gx, gy) __clang_ocl_kern_imp_meshTrace((uchar4*)pixels, c, (Ray*)rays, t_size, (float3*)pos, (float3*)nor, mindex, (float4*)mcolor,
                                                  b, n_slabs, slab_size);
}
void ref_a07_molTrace(void* pixels, const float* cam, void* rays, unsigned s_size, void* atoms, unsigned* mindex, void* mcolor,
                      const float* bound, unsigned n_slabs, unsigned* slab_size, size_t gx, size_t gy) {
    float16 c = mk16(cam);
    AABB b = mkbox(bound);
    FOR_2D(gx, gy) __clang_ocl_kern_imp_molTrace((uchar4*)pixels, c, (Ray*)rays, s_size, (float4*)atoms, mindex, (float4*)mcolor, b, n_slabs, slab_size);
}
}
#endif
