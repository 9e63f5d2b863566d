// pt_launch.hpp -- host-callable launchers of the kernel families (C++ linkage; the C ABI
// in mirt_abi.cpp is the only caller).  All launches are asynchronous on `s`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pt {

enum { KIND_SPHERES = 0, KIND_TRIANGLES = 1 };

// ---- reference-shaped kernels (pt_kernels_granular.hip) ------------------------------------
void launch_sizeof(hipStream_t s, bool ray, uint32_t* out);
void launch_initAcu(hipStream_t s, void* acu, uint32_t total, uint32_t gsz);
void launch_lensDraws(hipStream_t s, void* seeds, void* uv, uint32_t cols, uint32_t rows, uint32_t gx, uint32_t gy,
                      uint32_t row0, uint32_t nrows);
void launch_initTrace(hipStream_t s, void* rays, void* pois, const void* uv, const float* bound, const float* cam,
                      float focal, float lens_rad, uint32_t rpp, uint32_t gx, uint32_t gy);
void launch_bouncePaths(hipStream_t s, const void* pois, void* rays, void* seeds, uint32_t total, uint32_t gsz);
void launch_lightRender(hipStream_t s, void* pois, void* rays, void* acu, const float* light, uint32_t total, uint32_t gsz);
void launch_initShadowTrace(hipStream_t s, void* shadow, const void* pois, uint32_t total, const float* light, void* seeds, uint32_t gsz);
void launch_closest(hipStream_t s, int kind, uint32_t total, void* pois, void* rays, const void* prims, const void* normals,
                    const void* matid, uint32_t mesh_matid, const void* off, const float* bound, uint32_t n, uint32_t exit_far, uint32_t gsz);
void launch_anyhit(hipStream_t s, int kind, uint32_t total, void* shadow, const void* prims, const void* off, const float* bound,
                   uint32_t n, uint32_t exit_far, uint32_t gsz);
void launch_sceneRender(hipStream_t s, void* acu, void* pois, const void* shadow, const void* material, uint32_t nmat,
                        const float* light, uint32_t total, uint32_t gsz);
void launch_copyToPixel(hipStream_t s, void* pixel, const void* acu, float m, uint32_t pixels, uint32_t rpp, uint32_t gsz, void* radiance);
void launch_numerics(hipStream_t s, int op, const void* a, const void* b, void* out, uint64_t n);
void launch_divCheck(hipStream_t s, int mode, uint64_t seed, uint64_t count, void* out16);
void launch_seedFill(hipStream_t s, void* seeds, uint64_t first, uint64_t count, uint32_t base);

// ---- fused pass (pt_kernels_fused.hip) -------------------------------------------------------
constexpr int kMaxLights = 8;
constexpr int kMaxMeshes = 16;

constexpr uint32_t kNoLds = 0xFFFFFFFFu;
constexpr uint32_t kLdsOffWords = 4096;   // 16 KB of LDS per block for cell-offset tables (n <= 15 for a single grid)
constexpr uint32_t kLdsTriMax = 96;       // single-cell triangle sets staged in LDS for the per-lane candidate loops (records, vertex normals, material ids:
                                          // 112 B per triangle): 10.5 KB per block at most
struct GridArgs {            // one cell-sorted primitive set, device pointers
    const void* prims;       // float4 per sphere (c, r^2) | 3 x float4 per PREPARED triangle (launch_prepTriangles)
    const void* pnorm;       // triangles, at most kLdsTriMax records: the candidate sweep's PLANE LIST behind the records and group spheres
                             // (prepared_planes_offset; k_planeList writes it).  A 64-byte header, one slot per chunk of 32 records:
                             // {groups[4], first[4], Gmax[4], Hmax[4]} -- groups = 64-byte groups per class (x | y << 8 | z << 16 | general << 24),
                             // first = the chunk's first group, Gmax / Hmax = the chunk's largest margin constants (M = G |o|_1 + H).  Then the
                             // groups: axis planes {q, n_a, mask, back mask} four to a group, general planes {n.xyz, k = p0 . n, mask, back
                             // mask, 0, 0} two to a group; mask = the chunk's records in that plane (record c0 + j at bit 31 - j), back mask = those
                             // in the same plane with the normal reversed.  Read by scalar loads in the sweep (pt_trace.hpp trace_cell1).
                             // Null: no list (the set runs the wave-uniform loop)
    const void* normals;     // 3 x float4 per triangle (null for spheres)
    const void* matid;       // uint per primitive (null: use `mesh_matid`)
    const void* off;         // uint[n^3 + 1]
    float bound[8];          // (min,1,max,1)
    uint32_t n;              // cells per axis
    uint32_t mesh_matid;
    uint32_t kind;           // KIND_SPHERES | KIND_TRIANGLES
    uint32_t fast_ok;        // geometry-side guard of the exact 3-operation divisions (pt_trace.hpp ray_recip): every bound is 0 or
                             // in [2^-30, 2^20]; for triangles every plane-normal component is 0 or in [2^-40, 2^40]
    uint32_t lds_off;        // fused pass (launch_fused assigns it).  n > 1: dword index of this set's cell-offset table inside the block's
                             // LDS copy, or kNoLds when the tables of the scene do not fit.  n == 1, triangles: dword index (a multiple of
                             // 4) of the set's `nslots` prepared records staged in LDS for the per-lane candidate loops, or kNoLds
    float delta[3], rdelta[3]; // n > 1, optimistic kernel: the cell width per axis (hi - lo) / n and its reciprocal, both correctly rounded --
    uint32_t nslots;         // off[n^3], the number of (cell, primitive) slots, when the host knows it (0: the walk reads it from the table)
    uint32_t first_zero;     // off[0] == 0 (host-checked): a single-cell set's list is slots [0, nslots)
    uint32_t walk_ok;        // what every lane would compute for itself from wave-uniform inputs (pt_trace.hpp axis_setup_t).  walk_ok: the
                             // widths and spans sit inside the windows in which the kernel's 3-operation divisions are exact; a lane that
                             // walks a set without it hands its sample to the exact kernel
    uint32_t exit_is_far_face; // n == 1 only: lo + 1*((hi-lo)/1) == hi and lo + 0*((hi-lo)/1) == lo hold bitwise on all three
                             // axes (checked on the host), so the single cell's exit t equals the AABB slab's far t
};
struct LightArgs {           // the three float16 packings of one light (A10 code.js:323-352)
    float shadow[16];        // pos, T, B, radius
    float scene[16];         // pos, normal, irradiance, area
    float light[16];         // pos, normal, irradiance, radius
};
struct FusedArgs {
    float cam[16];
    float bound[8];
    float focal_length, lens_rad;
    uint32_t width, height, rpp;
    uint32_t row0, nrows;    // row tile this launch renders; per-ray buffers are tile-local
    uint32_t bounces;        // 5 in the reference (A10 code.js:1829)
    uint32_t n_sets, n_lights;
    GridArgs sets[2 + kMaxMeshes];   // upload order: spheres, loose triangles, mesh 0..M-1 (A10 code.js:1809-1813)
    LightArgs lights[kMaxLights];
    const void* material;    // float4[nmat]
    uint32_t nmat;
    int32_t* seeds;          // [nrows*width*rpp], read-modify-write
    void* acu;               // float4[nrows*width*rpp], accumulated into
    const void* uv;          // rpp == 1: float2[nrows*width] lens draws from launch_lensDraws
    uint32_t fresh;          // 1: the accumulator starts at zero and is not read (initAcu folded into the pass, mirt_render_first_pass)
    // copyToPixel INSIDE the pass (A10 code.cl:1366-1386; `resolve` != 0): a block of 256 consecutive ray ids holds whole pixels (rpp divides 256) and the
    // pass is a frame's first, so every accumulator of a pixel is final in the block's LDS when its last sample ends: the block sums them in the
    // reference's order and writes `pixel` (RGBA8) and / or `radiance` (the un-scaled sums); `acu` may then be null -- nothing per ray but the seed
    // touches memory: 8 B per sample + 20 B per pixel.  launch_fused decides (fused_resolves()).
    void* pixel;             // uchar4[nrows*width] or null
    void* radiance;          // float4[nrows*width] or null
    float res_m;             // 1 / (rpp * passes), A10 code.js:1412
    uint32_t resolve;
    // A pixel of MORE than 256 rays (rpp = 256 * chunks; chunks a power of two <= 32: 1024 rays are 4) resolves in `chunks` launches: launch c renders the
    // c-th block of 256 ray ids of EVERY pixel (workgroup b: block b * chunks + c) and continues the pixel's chain of additions from what `radiance`
    // holds -- the sums over the blocks before it, written by launches 0 .. c-1 -- so the reference's one chain (A10 code.cl:1377-1380) is cut at
    // multiples of 256 and carried through memory, 16 B per pixel and launch instead of 16 B per ray.  `pixel` is given to the last launch only.
    uint32_t chunks;         // 1, or rpp / 256
    uint32_t chunk;          // this launch's c
    uint32_t chunk_bits;     // redo mode: the bits of a 32-block mask word that are blocks of this launch (all ones when chunks == 1)
};
// whether a pass with these arguments resolves inside the kernel: whole pixels per block (or whole blocks per pixel, see FusedArgs::chunks) and
// somewhere to put the result.  A frame's first pass may then do without `acu`; a later pass reads and writes it as ever -- the block's LDS holds the
// accumulators as the pass leaves them, which is what the separate copyToPixel would read back.
inline uint32_t fused_chunks(uint32_t rpp) { return rpp > 256u ? rpp / 256u : 1u; }
inline bool fused_resolves(uint32_t rpp, bool want_out) {
    if (!want_out || rpp == 0u) return false;
    if (rpp <= 256u) return 256u % rpp == 0u;
    const uint32_t c = rpp / 256u;
    return rpp % 256u == 0u && c <= 32u && (c & (c - 1u)) == 0u;
}
// fast: the optimistic kernel (writes deferred samples' bits into defer_mask); !fast: the exact kernel over `list` (or everything)
void launch_fused(hipStream_t s, const FusedArgs& a, bool fast, uint32_t* defer_mask, const uint32_t* redo_mask, uint32_t redo_words);
bool fused_fast_available();   // compiled with PT_EXACT_FAST_DIV
void launch_deferCount(hipStream_t s, const uint32_t* mask, uint32_t words, uint32_t* count);
// {p0,e1,e2,n} records from the host's 3 x float4 position buffer (see pt_kernels_fused.hip); `out` holds count records of 48 B,
// behind them ceil(count / kTriGroup) float4 {centre, R'^2}: the bounding spheres of groups of consecutive records, and behind those, for
// count <= kLdsTriMax, the candidate sweep's plane list (GridArgs::pnorm)
void launch_prepTriangles(hipStream_t s, const void* pos, void* out, uint32_t count, uint32_t* insane_word);
size_t prepared_bytes(uint32_t count);   // what `out` must hold for `count` triangles
#ifndef PT_TRI_GROUP
#define PT_TRI_GROUP 16
#endif
constexpr uint32_t kTriGroup = PT_TRI_GROUP;   // prepared records per bounding sphere (pt_trace.hpp group_missed)
// byte offset of the candidate sweep's plane list inside it (64-byte aligned), and its size: the header, then per chunk of 32 records at
// most 32 general entries of 32 bytes (or 32 axis entries of 16) plus one partly filled 64-byte group per class
inline size_t prepared_planes_bytes(uint32_t count) { return 64 + ((size_t)count + 31) / 32 * (32 * 32 + 4 * 64); }
__host__ __device__ inline size_t prepared_planes_offset(uint32_t count) { return ((size_t)count * 48 + ((size_t)count + kTriGroup - 1) / kTriGroup * 16 + 63) & ~(size_t)63; }

// ---- uniform-grid build on the device (pt_grid_build.hip) -------------------------------------------
constexpr uint32_t kMaxCellSlots = 1u << 24;        // slots ONE cell of a grid with n > 1 may hold: the shared-test walk packs a slot's place in its cell into 24 bits (pt_trace_coop.hpp)
constexpr uint64_t kMaxGridSlots = 0x7FFFFFFFull;   // (cell, primitive) slots one grid may hold: the sort and every consumer index them with 31 bits
hipError_t grid_build(hipStream_t s, int kind, const double* prims, uint32_t count, const double bounds6[6], uint32_t n,
                      uint32_t* offsets, uint32_t** order_out, uint32_t* total, uint64_t* slots_needed);
void launch_gatherTriangles(hipStream_t s, const uint32_t* order, uint32_t total, const double* pos9, const double* nor9,
                            int nsteps, const int* ops, const double* vecs, float pad_w, void* pos_out, void* nor_out);
void launch_gatherSpheres(hipStream_t s, const uint32_t* order, uint32_t total, const double* sph4, void* out);
// parseMeshJSON for one (node, mesh) pair: corners de-indexed + transformed into the fp64 soups, bounds6 (fp32 min xyz, max xyz) merged.
// scratch8: 8 device words (6 encoded bounds, a flag word set to 1 by an out-of-range index, one spare); the caller zeroes word 6 first.
void launch_meshIngest(hipStream_t s, const double* P, const double* N, const uint32_t* idx, uint32_t n_vertices, uint32_t n_corners,
                       const float* m16, const float* nm9, double* pos_out, double* nor_out, float* bounds6, uint32_t* scratch8);
void launch_gatherU32(hipStream_t s, const uint32_t* order, uint32_t total, const uint32_t* in, uint32_t* out);

// ---- single-frame kernels of Assign01 / 04 / 07 (pt_kernels_frame.hip) -----------------------------
void launch_a01_raytrace(hipStream_t s, void* pixels, const float* cam, uint32_t gx, uint32_t gy);
void launch_frame_initTrace(hipStream_t s, bool clip, void* pixels, const float* cam, void* rays, const float* bound, uint32_t gx, uint32_t gy);
void launch_a04_meshTrace(hipStream_t s, void* pixels, const float* cam, void* rays, uint32_t t_size, const void* prep, const void* normals,
                          const void* mindex, const void* mcolor, uint32_t ncolors, uint32_t gx, uint32_t gy);
// prep: the prepared records of the n_slots grid slots followed by one bounding sphere per kTriGroup records (launch_prepTriangles)
void launch_a07_meshTrace(hipStream_t s, void* pixels, const float* cam, void* rays, const void* prep, const void* normals, const float* bound,
                          uint32_t n_slabs, const void* slab_size, uint32_t n_slots, uint32_t gx, uint32_t gy);
void launch_a07_molTrace(hipStream_t s, void* pixels, const float* cam, void* rays, const void* atoms, const float* bound, uint32_t n_slabs,
                         const void* slab_size, uint32_t gx, uint32_t gy);

}  // namespace pt
