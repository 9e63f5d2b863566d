// pt_grid_build.hip -- the uniform-grid builders of the reference host on the device.
//
// The reference bins primitives into an n x n x n grid on the CPU with nested JS arrays
// (splitSphereData / splitTriangleData / splitMeshData, A10 code.js:1554-1772, 899-1041):
//   lo = floor((boxmin - bmin) / w), hi = floor((boxmax - bmin) / w) per axis in DOUBLE precision,
//   lo clamped from below only, hi from above only (so a primitive on the max face has lo = n > hi = n-1
//   and is silently dropped), every cell of [lo,hi]^3 receives a copy; cells are emitted z-major, then y,
//   then x, primitives inside a cell in input order.
// The same result as a data-parallel pipeline (fp64 throughout, so every floor() sees the bits JS sees):
//   1. k_cellRanges   one thread per primitive: AABB from its fp64 vertices, the six clamped indices, #cells
//   2. exclusive scan of #cells (rocPRIM)            -> where each primitive's (cell, prim) pairs start
//   3. k_emitPairs    one thread per primitive: its pairs in z,y,x order; per-cell counts by atomics
//   4. stable radix sort of the pairs by cell (rocPRIM)   -> `order` (input order survives inside a cell)
//   5. exclusive scan of the per-cell counts              -> `offsets[n^3 + 1]`
// plus k_gatherTriangles / k_gatherSpheres: slot arrays (3 x float4 positions / normals, float4 spheres)
// from `order`, applying the mesh's normalise / scale / translate in fp64 before narrowing to fp32
// exactly where `new Float32Array(...)` narrows (A10 code.js:114-169, 1263-1265).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "pt_launch.hpp"

namespace pt {

struct GridDims { double bmin[3], w[3]; uint32_t n; };

__device__ __forceinline__ void clamp_range(const double* lo, const double* hi, const GridDims& g, int* r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a = floor((lo[c] - g.bmin[c]) / g.w[c]);
        double b = floor((hi[c] - g.bmin[c]) / g.w[c]);
        // JS: `if (a < 0) a = 0; if (b >= n) b = n-1;` on doubles (NaN compares false and then runs no loop iteration)
        if (a < 0) a = 0;
        if (b >= (double)g.n) b = (double)g.n - 1.0;
        // loop `for (i = a; i <= b; i++)`: empty when a > b or either is NaN; clamp the other side only to keep ints sane
        int ia = (a == a && a <= (double)g.n) ? (int)a : (int)g.n;        // a > n-1 -> empty anyway
        int ib = (b == b && b >= -1.0) ? (int)b : -1;
        r[c] = ia;
        r[3 + c] = ib;
    }
}

// kind 0: spheres, prim = (cx, cy, cz, r) fp64; kind 1: triangles, prim = 9 fp64 (p0, p1, p2)
__global__ void __launch_bounds__(256) k_cellRanges(int kind, const double* prims, uint32_t count, GridDims g, int* ranges, uint64_t* ncells) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double lo[3], hi[3];
    if (kind == 0) {
        const double* s = prims + 4u * (size_t)i;
        for (int c = 0; c < 3; ++c) { lo[c] = s[c] - s[3]; hi[c] = s[c] + s[3]; }
    } else {
        const double* p = prims + 9u * (size_t)i;
        for (int c = 0; c < 3; ++c) {
            // Math.min(Math.min(a,b),c): NaN-propagating in JS; fmin would drop NaNs, so spell the comparisons out
            double a = p[c], b = p[3 + c], d = p[6 + c];
            double mn = (a != a || b != b) ? (a + b) : (a < b ? a : b);
            mn = (mn != mn || d != d) ? (mn + d) : (mn < d ? mn : d);
            double mx = (a != a || b != b) ? (a + b) : (a > b ? a : b);
            mx = (mx != mx || d != d) ? (mx + d) : (mx > d ? mx : d);
            lo[c] = mn; hi[c] = mx;
        }
    }
    int r[6];
    clamp_range(lo, hi, g, r);
    uint64_t nc = 1;   // up to n^3 = 2^30 cells per primitive, times up to 2^32 primitives: counted in 64 bits, checked on the host
    for (int c = 0; c < 3; ++c) nc *= (r[3 + c] >= r[c]) ? (uint64_t)(r[3 + c] - r[c] + 1) : 0ull;
    for (int c = 0; c < 6; ++c) ranges[6u * (size_t)i + c] = r[c];
    ncells[i] = nc;
}

__global__ void __launch_bounds__(256) k_emitPairs(const int* ranges, const uint64_t* start, uint32_t count, uint32_t n,
                                                    uint32_t* keys, uint32_t* vals, uint32_t* cell_count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int* r = ranges + 6u * (size_t)i;
    uint64_t k = start[i];   // < 2^31 in total: grid_build refuses larger slot counts before this kernel is launched
    for (int z = r[2]; z <= r[5]; ++z)
        for (int y = r[1]; y <= r[4]; ++y)
            for (int x = r[0]; x <= r[3]; ++x) {
                const uint32_t cell = ((uint32_t)z * n + (uint32_t)y) * n + (uint32_t)x;
                keys[k] = cell;
                vals[k] = i;
                ++k;
                atomicAdd(&cell_count[cell], 1u);
            }
}

// up to four fp64 per-axis steps applied in order: 0 = subtract, 1 = multiply, 2 = add  (Mesh.normalize = sub centre,
// mul 1/maxdim; Mesh.scale = mul; Mesh.translate = add: A10 code.js:114-169)
struct Xform { int nsteps; int op[4]; double v[4][3]; };

__device__ __forceinline__ double apply_xf(double x, int c, const Xform& xf) {
    for (int s = 0; s < xf.nsteps; ++s) {
        if (xf.op[s] == 0) x = x - xf.v[s][c];
        else if (xf.op[s] == 1) x = x * xf.v[s][c];
        else x = x + xf.v[s][c];
    }
    return x;
}

__global__ void __launch_bounds__(256) k_gatherTriangles(const uint32_t* order, uint32_t total, const double* pos9, const double* nor9,
                                                          Xform xf, float pad_w, float4* pos_out, float4* nor_out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const uint32_t i = order[k];
    for (int v = 0; v < 3; ++v) {
        const double* p = pos9 + 9u * (size_t)i + 3 * v;
        pos_out[3u * (size_t)k + v] = make_float4((float)apply_xf(p[0], 0, xf), (float)apply_xf(p[1], 1, xf), (float)apply_xf(p[2], 2, xf), pad_w);
        if (nor_out) {
            const double* q = nor9 + 9u * (size_t)i + 3 * v;
            nor_out[3u * (size_t)k + v] = make_float4((float)q[0], (float)q[1], (float)q[2], 0.0f);
        }
    }
}
__global__ void __launch_bounds__(256) k_gatherSpheres(const uint32_t* order, uint32_t total, const double* sph4, float4* out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const double* s = sph4 + 4u * (size_t)order[k];
    out[k] = make_float4((float)s[0], (float)s[1], (float)s[2], (float)(s[3] * s[3]));   // rad*rad in fp64, then narrowed (code.js:1602)
}
__global__ void __launch_bounds__(256) k_gatherU32(const uint32_t* order, uint32_t total, const uint32_t* in, uint32_t* out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < total) out[k] = in[order[k]];
}

static inline dim3 g1(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// ---- mesh ingest: parseMeshJSON (A10 tri/meshDataVersion1.js:12-78) for one (node, mesh) pair -----------------------------------
// The reference walks nodes x meshes on the host: every vertex through the node's model matrix (gl-matrix vec3.transformMat4, :33-37,
// for the bounds), every triangle corner de-indexed and transformed again (:52-60), its normal through the normal matrix
// (vec3.transformMat3, :62-66).  gl-matrix keeps matrices and vectors in Float32Array: operands are the fp32 matrix entries widened to
// double, the sums are evaluated in double left to right (JavaScript never fuses), the result is rounded to fp32 on store.  Same here,
// one thread per corner; the outputs are the fp64 soups the grid builder consumes (every value fp32-representable).
struct MeshXf { float m[16]; float nm[9]; };
__device__ __forceinline__ uint32_t enc_f32(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b ^ 0x80000000u); }   // order-preserving
__device__ __forceinline__ float dec_f32(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e ^ 0x80000000u) : ~e); }
__device__ __forceinline__ void xf_point(const MeshXf& X, double x, double y, double z, float* o) {
    o[0] = (float)((double)X.m[0] * x + (double)X.m[4] * y + (double)X.m[8] * z + (double)X.m[12]);
    o[1] = (float)((double)X.m[1] * x + (double)X.m[5] * y + (double)X.m[9] * z + (double)X.m[13]);
    o[2] = (float)((double)X.m[2] * x + (double)X.m[6] * y + (double)X.m[10] * z + (double)X.m[14]);
}
__global__ void __launch_bounds__(256) k_meshCorners(const double* P, const double* N, const uint32_t* idx, uint32_t n_vertices, uint32_t n_corners,
                                                      MeshXf X, double* pos_out, double* nor_out, uint32_t* flags) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_corners) return;
    const uint32_t v = idx ? idx[c] : c;
    if (v >= n_vertices) { atomicOr(flags, 1u); return; }   // an index past the vertex array: reported, nothing read
    float o[3];
    xf_point(X, P[3u * (size_t)v], P[3u * (size_t)v + 1], P[3u * (size_t)v + 2], o);
    pos_out[3u * (size_t)c] = o[0]; pos_out[3u * (size_t)c + 1] = o[1]; pos_out[3u * (size_t)c + 2] = o[2];
    const double x = N[3u * (size_t)v], y = N[3u * (size_t)v + 1], z = N[3u * (size_t)v + 2];
    nor_out[3u * (size_t)c] = (float)(x * (double)X.nm[0] + y * (double)X.nm[3] + z * (double)X.nm[6]);
    nor_out[3u * (size_t)c + 1] = (float)(x * (double)X.nm[1] + y * (double)X.nm[4] + z * (double)X.nm[7]);
    nor_out[3u * (size_t)c + 2] = (float)(x * (double)X.nm[2] + y * (double)X.nm[5] + z * (double)X.nm[8]);
}
// bounds over ALL vertices of the mesh (:33-37), not only the indexed ones; `if (v < min) min = v`: a NaN changes nothing
__global__ void __launch_bounds__(256) k_meshBounds(const double* P, uint32_t n_vertices, MeshXf X, uint32_t* enc6) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vertices) return;
    float o[3];
    xf_point(X, P[3u * (size_t)v], P[3u * (size_t)v + 1], P[3u * (size_t)v + 2], o);
    for (int c = 0; c < 3; ++c)
        if (o[c] == o[c]) { atomicMin(&enc6[c], enc_f32(o[c])); atomicMax(&enc6[3 + c], enc_f32(o[c])); }
}
__global__ void k_boundsCode(float* bounds6, uint32_t* enc6, int decode) {
    const int c = threadIdx.x;
    if (c >= 6) return;
    if (decode) bounds6[c] = dec_f32(enc6[c]); else enc6[c] = enc_f32(bounds6[c]);
}
void launch_meshIngest(hipStream_t s, const double* P, const double* N, const uint32_t* idx, uint32_t n_vertices, uint32_t n_corners,
                       const float* m16, const float* nm9, double* pos_out, double* nor_out, float* bounds6, uint32_t* scratch8) {
    MeshXf X;
    for (int i = 0; i < 16; ++i) X.m[i] = m16[i];
    for (int i = 0; i < 9; ++i) X.nm[i] = nm9[i];
    hipLaunchKernelGGL(k_boundsCode, dim3(1), dim3(64), 0, s, bounds6, scratch8, 0);
    if (n_vertices) hipLaunchKernelGGL(k_meshBounds, g1(n_vertices), dim3(256), 0, s, P, n_vertices, X, scratch8);
    hipLaunchKernelGGL(k_boundsCode, dim3(1), dim3(64), 0, s, bounds6, scratch8, 1);
    if (n_corners) hipLaunchKernelGGL(k_meshCorners, g1(n_corners), dim3(256), 0, s, P, N, idx, n_vertices, n_corners, X, pos_out, nor_out, scratch8 + 6);
}

// Returns hipSuccess and the number of (cell, primitive) slots.  `offsets` must hold n^3 + 1 uints.  `order_out` receives
// a device allocation (hipMalloc) of `*total` uints that the caller owns.  `*slots_needed` always receives the exact 64-bit slot
// count; when it exceeds kMaxGridSlots nothing is emitted (order_out stays null, total 0) and the caller reports MIRT_E_RANGE.
// Device memory: one arena for the counting phase, one for the sort -- three hipMalloc per build including the result.
hipError_t grid_build(hipStream_t s, int kind, const double* prims, uint32_t count, const double bounds6[6], uint32_t n,
                      uint32_t* offsets, uint32_t** order_out, uint32_t* total, uint64_t* slots_needed) {
    GridDims g;
    g.n = n;
    for (int c = 0; c < 3; ++c) { g.bmin[c] = bounds6[c]; g.w[c] = (bounds6[3 + c] - bounds6[c]) / (double)n; }
    const uint64_t cells = (uint64_t)n * n * n;
    *order_out = nullptr;
    *total = 0;
    *slots_needed = 0;
    char *arena1 = nullptr, *arena2 = nullptr;
    hipError_t rc = hipSuccess;
    auto cleanup = [&]() { (void)hipFree(arena1); (void)hipFree(arena2); };
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
#define GB_TRY(e) do { rc = (e); if (rc != hipSuccess) { cleanup(); if (*order_out) { (void)hipFree(*order_out); *order_out = nullptr; } return rc; } } while (0)
    // ---- phase 1: per-primitive cell ranges and counts, their 64-bit scan, the per-cell counters
    size_t scan1 = 0, scan2 = 0;
    GB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan1, (uint64_t*)nullptr, (uint64_t*)nullptr, (int)count + 1, s));
    GB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan2, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)cells + 1, s));
    const size_t o_ranges = 0, o_ncells = o_ranges + up((size_t)count * 24), o_start = o_ncells + up(((size_t)count + 1) * 8),
                 o_cellcnt = o_start + up(((size_t)count + 1) * 8), o_tmp = o_cellcnt + up((cells + 1) * 4),
                 bytes1 = o_tmp + up(scan1 > scan2 ? scan1 : scan2);
    GB_TRY(hipMalloc((void**)&arena1, bytes1));
    int* ranges = (int*)(arena1 + o_ranges);
    uint64_t *ncells = (uint64_t*)(arena1 + o_ncells), *start = (uint64_t*)(arena1 + o_start);
    uint32_t* cell_count = (uint32_t*)(arena1 + o_cellcnt);
    void* tmp = arena1 + o_tmp;
    GB_TRY(hipMemsetAsync(cell_count, 0, (cells + 1) * 4, s));
    if (count) {
        GB_TRY(hipMemsetAsync(ncells, 0, ((size_t)count + 1) * 8, s));
        hipLaunchKernelGGL(k_cellRanges, g1(count), dim3(256), 0, s, kind, prims, count, g, ranges, ncells);
        size_t need = scan1;
        GB_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, need, ncells, start, (int)count + 1, s));
        uint64_t tot = 0;
        GB_TRY(hipMemcpyAsync(&tot, start + count, 8, hipMemcpyDeviceToHost, s));
        GB_TRY(hipStreamSynchronize(s));
        *slots_needed = tot;
        if (tot > kMaxGridSlots) { cleanup(); return hipSuccess; }   // the caller turns this into MIRT_E_RANGE; nothing was emitted
        *total = (uint32_t)tot;
        if (tot) {
            // ---- phase 2: (cell, primitive) pairs, stable sort by cell
            int bits = 1;
            while ((1ull << bits) < cells && bits < 32) ++bits;
            size_t sort_tmp = 0;
            GB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)tot, 0, bits, s));
            const size_t o_keys = 0, o_vals = o_keys + up((size_t)tot * 4), o_keys2 = o_vals + up((size_t)tot * 4), o_stmp = o_keys2 + up((size_t)tot * 4);
            GB_TRY(hipMalloc((void**)&arena2, o_stmp + up(sort_tmp)));
            GB_TRY(hipMalloc((void**)order_out, (size_t)tot * 4));
            uint32_t *keys = (uint32_t*)(arena2 + o_keys), *vals = (uint32_t*)(arena2 + o_vals), *keys2 = (uint32_t*)(arena2 + o_keys2);
            hipLaunchKernelGGL(k_emitPairs, g1(count), dim3(256), 0, s, ranges, start, count, n, keys, vals, cell_count);
            GB_TRY(hipcub::DeviceRadixSort::SortPairs(arena2 + o_stmp, sort_tmp, keys, keys2, vals, *order_out, (int)tot, 0, bits, s));
        }
    }
    size_t need = scan2;
    GB_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, need, cell_count, offsets, (int)cells + 1, s));
    GB_TRY(hipStreamSynchronize(s));
    cleanup();
    return hipSuccess;
#undef GB_TRY
}

void launch_gatherTriangles(hipStream_t s, const uint32_t* order, uint32_t total, const double* pos9, const double* nor9,
                            int nsteps, const int* ops, const double* vecs, float pad_w, void* pos_out, void* nor_out) {
    if (!total) return;
    Xform xf;
    xf.nsteps = nsteps;
    for (int i = 0; i < 4; ++i) { xf.op[i] = i < nsteps ? ops[i] : 0; for (int c = 0; c < 3; ++c) xf.v[i][c] = i < nsteps ? vecs[3 * i + c] : 0.0; }
    hipLaunchKernelGGL(k_gatherTriangles, g1(total), dim3(256), 0, s, order, total, pos9, nor9, xf, pad_w, (float4*)pos_out, (float4*)nor_out);
}
void launch_gatherSpheres(hipStream_t s, const uint32_t* order, uint32_t total, const double* sph4, void* out) {
    if (total) hipLaunchKernelGGL(k_gatherSpheres, g1(total), dim3(256), 0, s, order, total, sph4, (float4*)out);
}
void launch_gatherU32(hipStream_t s, const uint32_t* order, uint32_t total, const uint32_t* in, uint32_t* out) {
    if (total) hipLaunchKernelGGL(k_gatherU32, g1(total), dim3(256), 0, s, order, total, in, out);
}

}  // namespace pt
