/*
 * oracle/pt_oracle.h -- TEST INFRASTRUCTURE: CPU restatement of the Assign10
 * path-tracing kernels (and the A01/A04/A07 single-frame kernels), plain C.
 *
 * It is the CHECKER for the HIP path (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg).  Nothing under 2015-raytracing_amd/ includes,
 * links or calls it.  It is itself pinned against the reference's own OpenCL C
 * kernels compiled for x86 (oracle/_ref, build container only) through the
 * fixtures in tests/golden/ -- see oracle/README.md for what that pin does and
 * does not cover.
 *
 * Entry points mirror the reference kernels one-to-one; `gsz*` is the padded
 * global size the host would enqueue (work-items >= the real count return
 * early, as in the reference).  float16 arguments are const float[16], AABB
 * arguments const float[8] in the host packing (min,1,max,1).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float o[3], _p0, d[3], _p1, mint, maxt, _p2[2]; } pto_ray;        /* 48 B, code.cl:27-31 */
typedef struct { float p[3], _p0, n[3], _p1, atte[3], _p2; int32_t matId; int32_t _p3[3]; } pto_poi; /* 64 B, code.cl:57-62 */

unsigned oracle_a10_sizeofRay(void);
unsigned oracle_a10_sizeofPoi(void);
int oracle_a10_rand(int* seed);

void oracle_a10_initAcu(void* acu, unsigned total, size_t gsz);
void oracle_a10_initTrace(int* seeds, void* rays, void* pois, const float* bound, const float* cam,
                          float focal_length, float lens_rad, unsigned rpp, size_t gx, size_t gy);
void oracle_a10_bouncePaths(void* pois, void* rays, int* seeds, unsigned total, size_t gsz);
void oracle_a10_lightRender(void* pois, void* rays, void* acu, const float* light, unsigned total, size_t gsz);
void oracle_a10_initShadowTrace(void* shadow, void* pois, unsigned total, const float* light, int* seeds, size_t gsz);
void oracle_a10_sphereTrace(unsigned total, void* pois, void* rays, void* spheres, unsigned* matid,
                            unsigned* box, const float* bound, unsigned n, size_t gsz);
void oracle_a10_triangleTrace(unsigned total, void* pois, void* rays, void* pos, void* nor, unsigned* matid,
                              unsigned* box, const float* bound, unsigned n, size_t gsz);
void oracle_a10_meshTrace(unsigned total, void* pois, void* rays, void* pos, void* nor, unsigned* box,
                          unsigned matid, const float* bound, unsigned n, size_t gsz);
void oracle_a10_sphereShadowTrace(unsigned total, void* shadow, void* spheres, unsigned* box,
                                  const float* bound, unsigned n, size_t gsz);
void oracle_a10_triangleShadowTrace(unsigned total, void* shadow, void* pos, unsigned* box,
                                    const float* bound, unsigned n, size_t gsz);
void oracle_a10_sceneRender(void* acu, void* pois, void* shadow, void* material, const float* light,
                            unsigned total, size_t gsz);
void oracle_a10_copyToPixel(void* pixel, void* acu, float m, unsigned pixels, unsigned rpp, size_t gsz);

/* single-frame kernels of Assign01 / 04 / 07 (BASELINE configs 1-3) */
void oracle_a01_raytrace(void* pixels, const float* cam, size_t gx, size_t gy);
void oracle_a04_initTrace(void* pixels, const float* cam, void* rays, size_t gx, size_t gy);
void oracle_a04_meshTrace(void* pixels, const float* cam, void* rays, unsigned t_size, void* pos, void* nor, unsigned* mindex,
                          void* mcolor, size_t gx, size_t gy);
void oracle_a07_initTrace(void* pixels, const float* cam, void* rays, const float* bound, size_t gx, size_t gy);
void oracle_a07_meshTrace(void* pixels, const float* cam, void* rays, unsigned t_size, void* pos, void* nor, unsigned* mindex,
                          void* mcolor, const float* bound, unsigned n_slabs, unsigned* slab_size, size_t gx, size_t gy);
void oracle_a07_molTrace(void* pixels, const float* cam, void* rays, unsigned s_size, void* atoms, unsigned* mindex, void* mcolor,
                         const float* bound, unsigned n_slabs, unsigned* slab_size, size_t gx, size_t gy);

/* built-in probes (tests compare them with the functions the compiled reference called) */
float oracle_bi_sin(float x);
float oracle_bi_cos(float x);
/* the measured v_rsq_f32 / v_sqrt_f32 tables (oracle/hw_tables.bin.z, int8[2^24] each); must be set before the first kernel runs */
void oracle_set_hw_tables(const signed char* rsq_delta, const signed char* sqrt_delta);
/* the CPU model of one built-in over arrays; op = the kernel name of oracle/probe/builtins.cl without its "b_" */
void oracle_bi_eval(const char* op, const float* a, const float* b, const float* c, float* o, size_t n);

/* fp32 operation counter for the roofline's algorithmic-flop constant:
 * + - * / sqrt count 1 each, sin/cos count 1 each; compares, min/max, selects,
 * conversions and integer work count 0.  Only built with -DPTO_COUNT_FLOPS
 * (liboracle_count.so); single-threaded. */
unsigned long long oracle_flops_get(void);
void oracle_flops_reset(void);

/* threads used by the OpenMP loops (cpu_baseline reports it) */
int oracle_num_threads(void);
void oracle_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
