// pt_kernels_frame.hip -- the single-frame kernels of the reference's earlier assignments that
// BASELINE.json lists as parity configurations:
//   Assign01  raytrace                       (A01 code.cl:116-147)  one hard-coded sphere
//   Assign04  initTrace, meshTrace           (A04 code.cl:204-215, 262-315)  brute-force ray / triangle
//   Assign07  initTrace, meshTrace, molTrace (A07 code.cl:311-335, 475-626, 337-473)  3-D uniform grid DDA over triangles / atoms,
//                                            cell-parity shading
// Same launch shape as the reference (2-D NDRange, one work-item per pixel), same buffers
// (Ray 48 B, uchar4 pixels, float4-padded triangles).  Triangles are read from the prepared copy
// (pt_trace.hpp).  Assign04's molTrace (brute force over atoms) is not built: its molecule mode is subsumed by Assign07's.
#include "pt_trace.hpp"

namespace pt {

PT_DEV void store_ray48(RayAoS* p, const Ray& r) {
    float4* q = reinterpret_cast<float4*>(p);
    q[0] = make_float4(r.o.x, r.o.y, r.o.z, 0.0f);
    q[1] = make_float4(r.d.x, r.d.y, r.d.z, 0.0f);
    *reinterpret_cast<float2*>(q + 2) = make_float2(r.mint, r.maxt);
}
PT_DEV Ray load_ray48(const RayAoS* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 a = q[0], b = q[1];
    float2 c = *reinterpret_cast<const float2*>(q + 2);
    Ray r;
    r.o = mk3(a.x, a.y, a.z); r.d = mk3(b.x, b.y, b.z); r.mint = c.x; r.maxt = c.y;
    return r;
}
// getRay of A02..A10 (A04 code.cl:86-97): pinhole ray through the pixel centre
PT_DEV Ray pinhole_ray(const Cam& c, float col, float row) {
    float sx = (-0.5f + (col + 0.5f) / (float)c.cols) * c.width;
    float sy = (0.5f - (row + 0.5f) / (float)c.rows) * c.height;
    f3 cop = add3(fma3(sx, c.U, scl3(sy, c.V)), scl3(-1.0f, c.W));
    Ray r;
    r.d = norm3(cop);
    r.o = c.eye;
    r.mint = 0.0f;
    r.maxt = PT_INF;
    return r;
}
// (uchar) of a float the way the compiled reference does it: truncate to int32, keep the low byte
PT_DEV unsigned char f2u8(float f) { return (unsigned char)(f2i(f) & 0xFF); }

// ---- Assign01 -----------------------------------------------------------------------------------
// Camera packs rows, cols in .sE, .sF as floats (A01 code.cl:44-46; swapped w.r.t. A02+), getRay uses
// normalize(cop - eye), interSphere is the textbook quadratic with `/ 2*a` (sic) and an open interval.
__global__ void __launch_bounds__(256) k_a01_raytrace(uchar4* pixels, F16 cam, uint32_t gx, uint32_t gy) {
    uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t row = blockIdx.y * blockDim.y + threadIdx.y;
    const float rows = cam.v[14], cols = cam.v[15];
    // the reference has no range check (it is launched on exactly cols x rows); we add one so a padded
    // NDRange cannot write outside the image
    if (col >= gx || row >= gy || !((float)col < cols) || !((float)row < rows)) return;
    f3 eye = ld3(cam.v), U = ld3(cam.v + 3), V = ld3(cam.v + 6), W = ld3(cam.v + 9);
    float sx = (-0.5f + ((float)col + 0.5f) / cols) * cam.v[12];
    float sy = (0.5f - ((float)row + 0.5f) / rows) * cam.v[13];
    f3 cop = add3(fma3(sx, U, scl3(sy, V)), scl3(-1.0f, W));
    f3 o = eye;
    f3 d = norm3(sub3(cop, o));
    const f3 sc = mk3(0.0f, 0.0f, 1.0f);
    const float sr = 0.5f;
    f3 omc = sub3(o, sc);
    float a = dot3(d, d);
    float b = 2.0f * dot3(omc, d);
    float c = cl_fma(-sr, sr, dot3(omc, omc));              // A01 code.cl:68, contracted
    float dis = cl_fma(b, b, -((4.0f * a) * c));            // A01 code.cl:69: the LEFT product fuses
    bool v = false;
    float t = PT_INF;
    if (!(dis < 0.0f)) {
        float sq = cl_sqrt(dis);
        float t0 = (-b - sq) / 2 * a;
        float t1 = (-b + sq) / 2 * a;
        if (t0 > 0.0f && t0 < PT_INF) { t = t0; v = true; }
        else if (t1 > 0.0f && t1 < PT_INF) { t = t1; v = true; }
    }
    unsigned char base = v ? f2u8((1.0f - t) * 255.0f) : 0;
    pixels[f2u_uniform(cols) * row + col] = make_uchar4(base, base, base, 255);
}

// ---- Assign04 / Assign07 initTrace ---------------------------------------------------------------
template <bool CLIP>
__global__ void __launch_bounds__(256) k_frame_initTrace(uchar4* pixels, F16 cam16, RayAoS* rays, Box8 bound8, uint32_t gx, uint32_t gy) {
    const Cam cam = mk_cam(cam16);
    uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t row = blockIdx.y * blockDim.y + threadIdx.y;
    if (col >= gx || row >= gy || col >= cam.cols || row >= cam.rows) return;
    Ray r = pinhole_ray(cam, (float)col, (float)row);
    if (CLIP) clip_to(r, mk_box(bound8));   // A07 code.cl:321-328
    store_ray48(&rays[(size_t)cam.cols * row + col], r);
    pixels[(size_t)cam.cols * row + col] = make_uchar4(0, 0, 0, 255);
}

PT_DEV f3 interp_normal(const float4* normals, uint32_t i, float beta, float gamma) {
    const float4* nn = normals + 3u * (size_t)i;
    float w = 1.0f - beta - gamma;
    return norm3(fma3(gamma, ld3(nn[2]), fma3(w, ld3(nn[0]), scl3(beta, ld3(nn[1])))));
}

// ---- Assign04 meshTrace: every pixel against every triangle, wave-uniform loop --------------------
__global__ void __launch_bounds__(256) k_a04_meshTrace(uchar4* pixels, F16 cam16, RayAoS* rays, uint32_t t_size, const float4* prep,
                                                        const float4* normals, const uint32_t* mindex, const float4* mcolor,
                                                        uint32_t ncolors, uint32_t gx, uint32_t gy) {
    const Cam cam = mk_cam(cam16);
    uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t row = blockIdx.y * blockDim.y + threadIdx.y;
    if (col >= gx || row >= gy || col >= cam.cols || row >= cam.rows) return;
    const size_t pix = (size_t)cam.cols * row + col;
    Ray ray = load_ray48(&rays[pix]);
    float champ_t = PT_INF, cb = 0.0f, cg = 0.0f;
    uint32_t champ_i = t_size;
    // interTriangle (A04 code.cl:146-184) on the prepared triangle, staged: the 64 lanes of a wave are 2 x 32 adjacent pixels, and
    // almost every one of a mesh's small triangles is missed by all of them at an early rejection.  After each of the reference's
    // early-outs a wave ballot asks whether ANY lane is still in; if none is, the rest of the test is skipped for the wave.  Lanes
    // that are still in compute exactly the values the straight-line test computes, so nothing changes but the work.
    const float4* __restrict__ p = prep;
    const float4* __restrict__ groups = prep + 3u * (size_t)t_size;   // one bounding sphere per kTriGroup records (k_prepTriangles)
    const float dd = ray.d.x * ray.d.x + ray.d.y * ray.d.y + ray.d.z * ray.d.z;
    for (uint32_t i = 0; i < t_size; ++i, p += 3) {
        // a whole group none of the wave's 64 rays can reach (pt_trace.hpp group_missed) is skipped
        if ((i % kTriGroup) == 0u && i + kTriGroup <= t_size && __builtin_amdgcn_ballot_w64(!group_missed(ray, dd, groups[i / kTriGroup])) == 0ull) {
            i += kTriGroup - 1u;
            p += 3u * (kTriGroup - 1u);
            continue;
        }
        const float4 A = p[0], B = p[1], C = p[2];
        const f3 p0 = mk3(A.x, A.y, A.z), e1 = mk3(B.x, B.y, B.z), e2 = mk3(C.x, C.y, C.z), n = mk3(A.w, B.w, C.w);
        const float div = dot3(n, ray.d);
        bool in = !(div <= 0);                                                   // code.cl:153
        if (__builtin_amdgcn_ballot_w64(in) == 0ull) continue;
        const float idiv = rcp_exact(div, !in);                                  // 1.0f / div, bit for bit (pt_numerics.hpp)
        const f3 s = sub3(ray.o, p0);
        const float beta = dot3(cross3(s, ray.d), e2) * idiv;
        in = in & !(beta < 0.0f) & !(beta > 1.0f);                               // :161
        if (__builtin_amdgcn_ballot_w64(in) == 0ull) continue;
        const float gamma = dot3(cross3(s, e1), ray.d) * idiv;
        const float gb = gamma + beta;
        in = in & !(gamma < 0.0f) & !(gamma > 1.0f) & !(gb < 0.0f) & !(gb > 1.0f);   // :169-170 (A04 also rejects gamma > 1)
        if (__builtin_amdgcn_ballot_w64(in) == 0ull) continue;
        const float t = dot3(cross3(s, e2), e1) * -idiv;
        in = in & (t > ray.mint) & (t < ray.maxt);                               // open interval, :177
        if (in && t < champ_t) { champ_t = t; champ_i = i; cb = beta; cg = gamma; }
    }
    if (champ_i >= t_size) return;
    rays[pix].maxt = champ_t;
    f3 n = interp_normal(normals, champ_i, cb, cg);
    float shade = cl_clamp(dot3(cam.W, n), 0.0f, 1.0f);
    uint32_t m = mindex[champ_i];
    if (m >= ncolors) return;  // foreign-memory guard (the reference would read out of bounds)
    float4 mc = mcolor[m];
    pixels[pix] = make_uchar4(f2u8((mc.x * 255.0f) * shade), f2u8((mc.y * 255.0f) * shade), f2u8((mc.z * 255.0f) * shade), 255);
}

// ---- Assign07 meshTrace: 3-D grid DDA, colour = parity of the hit cell x fake shade -------------------
// GROUPS: coarse grids (see the loop); a template parameter so that fine grids run the plain loop, untouched
template <bool GROUPS>
__global__ void __launch_bounds__(256) k_a07_meshTrace(uchar4* pixels, F16 cam16, RayAoS* rays, const float4* prep, const float4* normals,
                                                        Box8 bound8, uint32_t n_slabs, const uint32_t* slab_size, uint32_t group_slots, uint32_t gx, uint32_t gy) {
    const Cam cam = mk_cam(cam16);
    uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t row = blockIdx.y * blockDim.y + threadIdx.y;
    if (col >= gx || row >= gy || col >= cam.cols || row >= cam.rows) return;
    const size_t pix = (size_t)cam.cols * row + col;
    Ray ray = load_ray48(&rays[pix]);
    if (ray.mint == ray.maxt) return;
    const Box bound = mk_box(bound8);
    BoxHit bh = inter_aabb(ray, bound);
    if (!bh.v) return;
    // per-lane DDA; the hit cell is needed for the colour, so the walk is spelled out here
    Axis ax = axis_setup(ray.o.x, ray.d.x, bh.tmin, bound.lo.x, bound.hi.x, n_slabs);
    Axis ay = axis_setup(ray.o.y, ray.d.y, bh.tmin, bound.lo.y, bound.hi.y, n_slabs);
    Axis az = axis_setup(ray.o.z, ray.d.z, bh.tmin, bound.lo.z, bound.hi.z, n_slabs);
    float champ_t = ray.maxt, cb = 0.0f, cg = 0.0f;
    uint32_t champ_i = UINT32_MAX;
    int hx = 0, hy = 0, hz = 0;
    const uint32_t zs = n_slabs * n_slabs, ys = n_slabs;
    // phase A / phase B as in trace_dda (pt_trace.hpp): close and open cells until every live lane holds a triangle, then one test each
    float t = bh.tmin, cmin = t, cmax = cl_min(cl_min(ax.tnext, ay.tnext), az.tnext);
    uint32_t cell = __umul24((uint32_t)az.slab, zs) + __umul24((uint32_t)ay.slab, ys) + (uint32_t)ax.slab;
    uint32_t i = slab_size[cell], end = slab_size[cell + 1];
    const float dd = GROUPS ? ray.d.x * ray.d.x + ray.d.y * ray.d.y + ray.d.z * ray.d.z : 0.0f;
    for (;;) {
        bool alive = true;
        while (i == end) {
            if (champ_i != UINT32_MAX) { alive = false; break; }
            t = cmax;
            if (t == ax.tnext) {
                ax.tnext += ax.dt;
                ax.slab += ax.dslab;
                if (t >= bh.tmax || ax.slab == ax.limit) { alive = false; break; }
            } else if (t == ay.tnext) {
                ay.tnext += ay.dt;
                ay.slab += ay.dslab;
                if (t >= bh.tmax || ay.slab == ay.limit) { alive = false; break; }
            } else {
                az.tnext += az.dt;
                az.slab += az.dslab;
                if (t >= bh.tmax || az.slab == az.limit) { alive = false; break; }
            }
            cmin = t;
            cmax = cl_min(cl_min(ax.tnext, ay.tnext), az.tnext);
            cell = __umul24((uint32_t)az.slab, zs) + __umul24((uint32_t)ay.slab, ys) + (uint32_t)ax.slab;
            i = slab_size[cell];
            end = slab_size[cell + 1];
        }
        if (!alive) break;
        // Coarse grids (the page's n_slabs = 2: a thousand slots per cell): when every lane still in has at least a whole group of
        // kTriGroup slots ahead in its list and none of their rays can reach its own group's bounding sphere (pt_trace.hpp group_missed),
        // all step over that group together -- per lane the skip is exact on its own, the ballot only keeps the wave in step.
        if (GROUPS) {
            const bool whole = ((i % kTriGroup) == 0u) & (end - i >= kTriGroup);
            if (__builtin_amdgcn_ballot_w64(!whole) == 0ull &&
                __builtin_amdgcn_ballot_w64(!group_missed(ray, dd, (prep + 3u * (size_t)group_slots)[i / kTriGroup])) == 0ull) {
                i += kTriGroup;
                continue;
            }
        }
        float ti, b, g;
        const float4* __restrict__ p = prep + 3u * (size_t)i;
        // adjacent pixels walk the same cells in nearly the same order: staged rejection with wave ballots (tri_test_staged)
        const bool hit = tri_test_staged<TRI_A07>(true, ray.o, ray.d, cmin, cmax, p[0], p[1], p[2], ti, b, g);
        if (hit && ti < champ_t) { champ_t = ti; champ_i = i; cb = b; cg = g; hx = ax.slab; hy = ay.slab; hz = az.slab; }
        ++i;
    }
    if (champ_i == UINT32_MAX) return;
    rays[pix].maxt = champ_t;
    f3 n = interp_normal(normals, champ_i, cb, cg);
    float shade = cl_clamp(dot3(cam.W, n), 0.0f, 1.0f);
    float k = shade * 127.0f;                              // A07 code.cl:616-622
    pixels[pix] = make_uchar4(f2u8((float)((hx % 2) + 1) * k), f2u8((float)((hy % 2) + 1) * k), f2u8((float)((hz % 2) + 1) * k), 255);
}

// ---- Assign07 molTrace (code.cl:337-473): the same grid walk over atoms {c, r*r}; colour = parity of the hit cell x fake shade.
// The hit record keeps the CELL of the champion (champ_slab, code.cl:402, 425-429): the walk ends once any cell produced one.
__global__ void __launch_bounds__(256) k_a07_molTrace(uchar4* pixels, F16 cam16, RayAoS* rays, const float4* atoms, Box8 bound8, uint32_t n_slabs,
                                                       const uint32_t* slab_size, uint32_t gx, uint32_t gy) {
    const Cam cam = mk_cam(cam16);
    uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t row = blockIdx.y * blockDim.y + threadIdx.y;
    if (col >= gx || row >= gy || col >= cam.cols || row >= cam.rows) return;
    const size_t pix = (size_t)cam.cols * row + col;
    Ray ray = load_ray48(&rays[pix]);
    if (ray.mint == ray.maxt) return;
    const Box bound = mk_box(bound8);
    BoxHit bh = inter_aabb(ray, bound);
    if (!bh.v) return;
    Axis ax = axis_setup(ray.o.x, ray.d.x, bh.tmin, bound.lo.x, bound.hi.x, n_slabs);
    Axis ay = axis_setup(ray.o.y, ray.d.y, bh.tmin, bound.lo.y, bound.hi.y, n_slabs);
    Axis az = axis_setup(ray.o.z, ray.d.z, bh.tmin, bound.lo.z, bound.hi.z, n_slabs);
    const SphereRay sr = sphere_ray<false>(ray.d);
    float champ_t = ray.maxt;
    uint32_t champ_i = UINT32_MAX;
    int hx = 0, hy = 0, hz = 0;
    const uint32_t zs = n_slabs * n_slabs, ys = n_slabs;
    // phase A / phase B as in trace_dda (pt_trace.hpp): close and open cells until every live lane holds an atom, then one test each
    float t = bh.tmin, cmin = t, cmax = cl_min(cl_min(ax.tnext, ay.tnext), az.tnext);
    uint32_t cell = __umul24((uint32_t)az.slab, zs) + __umul24((uint32_t)ay.slab, ys) + (uint32_t)ax.slab;
    uint32_t i = slab_size[cell], end = slab_size[cell + 1];
    for (;;) {
        bool alive = true;
        while (i == end) {
            if (champ_i != UINT32_MAX) { alive = false; break; }
            t = cmax;
            if (t == ax.tnext) {
                ax.tnext += ax.dt;
                ax.slab += ax.dslab;
                if (t >= bh.tmax || ax.slab == ax.limit) { alive = false; break; }
            } else if (t == ay.tnext) {
                ay.tnext += ay.dt;
                ay.slab += ay.dslab;
                if (t >= bh.tmax || ay.slab == ay.limit) { alive = false; break; }
            } else {
                az.tnext += az.dt;
                az.slab += az.dslab;
                if (t >= bh.tmax || az.slab == az.limit) { alive = false; break; }
            }
            cmin = t;
            cmax = cl_min(cl_min(ax.tnext, ay.tnext), az.tnext);
            cell = __umul24((uint32_t)az.slab, zs) + __umul24((uint32_t)ay.slab, ys) + (uint32_t)ax.slab;
            i = slab_size[cell];
            end = slab_size[cell + 1];
        }
        if (!alive) break;
        float ti;
        const bool hit = sph_test(ray.o, ray.d, sr, cmin, cmax, atoms[i], ti);
        if (hit && ti < champ_t) { champ_t = ti; champ_i = i; hx = ax.slab; hy = ay.slab; hz = az.slab; }
        ++i;
    }
    if (champ_i == UINT32_MAX) return;
    rays[pix].maxt = champ_t;
    const f3 ip = fma3(champ_t, ray.d, ray.o);
    const float shade = cl_clamp(dot3(cam.W, norm3(sub3(ip, ld3(atoms[champ_i])))), 0.0f, 1.0f);   // code.cl:455-457
    const float k = shade * 127.0f;                                                                // code.cl:463-469
    pixels[pix] = make_uchar4(f2u8((float)((hx % 2) + 1) * k), f2u8((float)((hy % 2) + 1) * k), f2u8((float)((hz % 2) + 1) * k), 255);
}

static F16 mk16f(const float* f) { F16 r; for (int i = 0; i < 16; ++i) r.v[i] = f[i]; return r; }
static Box8 mk8f(const float* f) { Box8 r; for (int i = 0; i < 8; ++i) r.v[i] = f ? f[i] : 0.0f; return r; }
static dim3 grid2(uint32_t gx, uint32_t gy) { return dim3((gx + 31) / 32, (gy + 7) / 8); }

void launch_a01_raytrace(hipStream_t s, void* pixels, const float* cam, uint32_t gx, uint32_t gy) {
    if (!gx || !gy) return;
    hipLaunchKernelGGL(k_a01_raytrace, grid2(gx, gy), dim3(32, 8), 0, s, (uchar4*)pixels, mk16f(cam), gx, gy);
}
void launch_frame_initTrace(hipStream_t s, bool clip, void* pixels, const float* cam, void* rays, const float* bound, uint32_t gx, uint32_t gy) {
    if (!gx || !gy) return;
    if (clip) hipLaunchKernelGGL(k_frame_initTrace<true>, grid2(gx, gy), dim3(32, 8), 0, s, (uchar4*)pixels, mk16f(cam), (RayAoS*)rays, mk8f(bound), gx, gy);
    else hipLaunchKernelGGL(k_frame_initTrace<false>, grid2(gx, gy), dim3(32, 8), 0, s, (uchar4*)pixels, mk16f(cam), (RayAoS*)rays, mk8f(nullptr), gx, gy);
}
void launch_a04_meshTrace(hipStream_t s, void* pixels, const float* cam, void* rays, uint32_t t_size, const void* prep, const void* normals,
                          const void* mindex, const void* mcolor, uint32_t ncolors, uint32_t gx, uint32_t gy) {
    if (!gx || !gy) return;
    hipLaunchKernelGGL(k_a04_meshTrace, grid2(gx, gy), dim3(32, 8), 0, s, (uchar4*)pixels, mk16f(cam), (RayAoS*)rays, t_size, (const float4*)prep,
                       (const float4*)normals, (const uint32_t*)mindex, (const float4*)mcolor, ncolors, gx, gy);
}
void launch_a07_meshTrace(hipStream_t s, void* pixels, const float* cam, void* rays, const void* prep, const void* normals, const float* bound,
                          uint32_t n_slabs, const void* slab_size, uint32_t n_slots, uint32_t gx, uint32_t gy) {
    if (!gx || !gy) return;
    // the group spheres behind the n_slots records are consulted only where cells are long on average (coarse grids)
    const uint64_t cells = (uint64_t)n_slabs * n_slabs * n_slabs;
    const uint32_t group_slots = (uint64_t)n_slots >= cells * 4u * kTriGroup ? n_slots : 0u;
    if (group_slots)
        hipLaunchKernelGGL(k_a07_meshTrace<true>, grid2(gx, gy), dim3(32, 8), 0, s, (uchar4*)pixels, mk16f(cam), (RayAoS*)rays, (const float4*)prep,
                           (const float4*)normals, mk8f(bound), n_slabs, (const uint32_t*)slab_size, group_slots, gx, gy);
    else
        hipLaunchKernelGGL(k_a07_meshTrace<false>, grid2(gx, gy), dim3(32, 8), 0, s, (uchar4*)pixels, mk16f(cam), (RayAoS*)rays, (const float4*)prep,
                           (const float4*)normals, mk8f(bound), n_slabs, (const uint32_t*)slab_size, 0u, gx, gy);
}

void launch_a07_molTrace(hipStream_t s, void* pixels, const float* cam, void* rays, const void* atoms, const float* bound, uint32_t n_slabs,
                         const void* slab_size, uint32_t gx, uint32_t gy) {
    if (!gx || !gy) return;
    hipLaunchKernelGGL(k_a07_molTrace, grid2(gx, gy), dim3(32, 8), 0, s, (uchar4*)pixels, mk16f(cam), (RayAoS*)rays, (const float4*)atoms, mk8f(bound),
                       n_slabs, (const uint32_t*)slab_size, gx, gy);
}

}  // namespace pt
