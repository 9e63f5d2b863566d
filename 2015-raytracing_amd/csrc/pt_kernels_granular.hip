// pt_kernels_granular.hip -- the fourteen kernels of A10 code.cl as HIP kernels for gfx950,
// same names, same argument lists, same buffer layouts (AoS Ray 48 B / Poi 64 B), so that
// the reference's own host code can drive them through the WebCL-shaped boundary
// (include/mirt.h).  One work-item per ray (or per pixel for initTrace / copyToPixel).
// These are HBM-bound by construction (every stage round-trips the ray state through
// memory, ~4.2 KB/sample on cornell.xml); the fast path is pt_kernels_fused.hip.
#include "pt_trace_coop.hpp"

namespace pt {

PT_DEV Ray load_ray(const RayAoS* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 a = q[0], b = q[1];
    float2 c = *reinterpret_cast<const float2*>(q + 2);
    Ray r;
    r.o = mk3(a.x, a.y, a.z);
    r.d = mk3(b.x, b.y, b.z);
    r.mint = c.x;
    r.maxt = c.y;
    return r;
}
PT_DEV void store_ray(RayAoS* p, const Ray& r) {
    float4* q = reinterpret_cast<float4*>(p);
    q[0] = make_float4(r.o.x, r.o.y, r.o.z, 0.0f);
    q[1] = make_float4(r.d.x, r.d.y, r.d.z, 0.0f);
    *reinterpret_cast<float2*>(q + 2) = make_float2(r.mint, r.maxt);
}
PT_DEV void store_dead(RayAoS* p) {  // code.cl:595, 647: only mint/maxt are defined
    *reinterpret_cast<float2*>(reinterpret_cast<float4*>(p) + 2) = make_float2(PT_INF, PT_INF);
}


// ---- block-coalesced AoS output --------------------------------------------------------------------------------------------
// The reference's work-item stores into its own Ray (48 B) / Poi (64 B): 16- and 8-byte pieces at a 48- or 64-byte lane stride,
// and often only one field of the struct (maxt; atte; mint+maxt).  Measured on MI355X that costs a third of the kernel
// (k_initShadowTrace 4.25 ms per 133 M rays as written by the work-items vs 2.91 ms below): partial sectors and three times the
// write transactions.  Here the block's 256 consecutive structs -- one contiguous run of float4s -- are assembled in LDS and
// written with unit-stride 16-byte stores.  `parts[struct]` says which of its float4s hold defined data (bit j = float4 j):
// a struct the work-item would not have touched is not written, a field it would have left alone is either skipped (its float4
// is masked out) or rewritten with the bits just loaded from it.  Pad words (.w of the float3s, the three ints after matId)
// are written as zero; nothing reads them.  Used where a kernel WRITES whole rays (initTrace, bouncePaths, initShadowTrace, the
// any-hit kernels, which now store nothing for a free ray).  Not used by k_closest (VALU-bound: 5.44 vs 5.51 ms) nor by
// k_sceneRender (rewriting the whole 64-byte vertex instead of its 12-byte atte: 5.06 -> 5.49 ms).
template <int F4>
PT_DEV void flush_structs(const float4* lds, const uint8_t* parts, void* base, uint32_t first, uint32_t lim) {
    __syncthreads();
    const uint32_t count = first < lim ? (lim - first < 256u ? lim - first : 256u) : 0u;
    float4* g = reinterpret_cast<float4*>(base) + (size_t)first * F4;
    for (uint32_t k = threadIdx.x; k < count * F4; k += 256u)
        if ((parts[k / F4] >> (k % F4)) & 1u) g[k] = lds[k];
}
PT_DEV void put_ray(float4* lds, const Ray& r) {
    lds[3u * threadIdx.x] = make_float4(r.o.x, r.o.y, r.o.z, 0.0f);
    lds[3u * threadIdx.x + 1] = make_float4(r.d.x, r.d.y, r.d.z, 0.0f);
    lds[3u * threadIdx.x + 2] = make_float4(r.mint, r.maxt, 0.0f, 0.0f);
}
PT_DEV void put_dead(float4* lds) { lds[3u * threadIdx.x + 2] = make_float4(PT_INF, PT_INF, 0.0f, 0.0f); }  // code.cl:595, 647: mint / maxt only
constexpr uint8_t kRayAll = 7, kRayTail = 4;          // parts masks: the whole Ray / only {mint, maxt}
constexpr uint8_t kPoiReset = 12;                  // Poi: atte + matId (initTrace)

// code.cl:440-446
__global__ void k_sizeofRay(uint32_t* out) { if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (uint32_t)sizeof(RayAoS); }
__global__ void k_sizeofPoi(uint32_t* out) { if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (uint32_t)sizeof(PoiAoS); }

// code.cl:448-456
__global__ void __launch_bounds__(256) k_initAcu(float4* acu, uint32_t total, uint32_t gsz) {
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= gsz || id >= total) return;
    acu[id] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// rpp == 1 only.  The reference draws the two lens coordinates from seeds[get_global_id(0)]
// inside a 2-D launch (code.cl:429 vs 469-470, 513-515): every row of a column shares one
// stream, and the order is a data race on any parallel device.  The oracle's definition
// (work-items in row-major order) makes row r of column c consume draws 2r+1, 2r+2; this
// pre-pass walks each column's stream serially, one thread per column, and leaves the
// coordinates in `uv` for k_initTrace.
__global__ void __launch_bounds__(64) k_lensDraws(int32_t* seeds, float2* uv, uint32_t cols, uint32_t rows,
                                                   uint32_t gx, uint32_t gy, uint32_t row0, uint32_t nrows) {
    uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= cols || col >= gx) return;
    int32_t s = seeds[col];
    uint32_t rmax = rows < gy ? rows : gy;
    for (uint32_t row = 0; row < rmax; ++row) {
        float cy = get_rand(s);
        float cx = get_rand(s);
        if (row >= row0 && row < row0 + nrows) uv[(size_t)(row - row0) * cols + col] = make_float2(cx, cy);
    }
    seeds[col] = s;
}

// code.cl:458-543.  The reference runs one work-item per pixel, which writes its rays_per_pixel rays and vertices one after the
// other (neighbouring lanes rpp x 48 bytes apart).  Here: one thread per RAY, ids in buffer order, so a block's output is one
// contiguous run.  The lens coordinates of sample (i, j) are rebuilt by the same repeated additions (coord += delta).
// `wc` / `wr`: columns / rows the reference's NDRange would have covered (min(global size, image size)).
__global__ void __launch_bounds__(256) k_initTrace(RayAoS* rays, PoiAoS* pois, const float2* uv, Box8 bound8, F16 cam16,
                                                    float focal_length, float lens_rad, uint32_t rpp, uint32_t wc, uint32_t wr) {
    __shared__ float4 sray[256 * 3];
    __shared__ float4 spoi[256 * 4];
    __shared__ uint8_t pr[256], pq[256];
    const Cam cam = mk_cam(cam16);
    const uint64_t n_rays = (uint64_t)cam.cols * wr * rpp;
    const uint64_t id = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint8_t mr = 0, mq = 0;
    if (id < n_rays) {
        const uint64_t pix = id / rpp;
        const uint32_t smp = (uint32_t)(id - pix * rpp);
        const uint32_t row = (uint32_t)(pix / cam.cols), col = (uint32_t)(pix - (uint64_t)row * cam.cols);
        if (col < wc) {
            const Box bound = mk_box(bound8);
            f3 fp = focal_point(cam, (float)col, (float)row, focal_length);
            bool have = true;
            float cx, cy;
            if (rpp > 1) {
                const uint32_t side = f2u_uniform(cl_sqrt((float)rpp));
                have = smp < side * side;             // rpp that is not a square: the k x k loops leave the tail rays unwritten
                const float delta = 1.0f / (float)side;
                const uint32_t i = side ? smp / side : 0u, j = smp - i * side;
                cy = delta / 2.0f;
                for (uint32_t k = 0; k < i; ++k) cy += delta;
                cx = delta / 2.0f;
                for (uint32_t k = 0; k < j; ++k) cx += delta;
            } else {
                float2 c = uv[(size_t)row * cam.cols + col];
                cx = c.x;
                cy = c.y;
            }
            if (have) {
                Ray r = thin_lens_ray(cam, fp, lens_rad, cx, cy);
                clip_to(r, bound);
                put_ray(sray, r);
                mr = kRayAll;
            }
            spoi[4u * threadIdx.x + 2] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
            spoi[4u * threadIdx.x + 3] = make_float4(__int_as_float(-1), 0.0f, 0.0f, 0.0f);
            mq = kPoiReset;
        }
    }
    pr[threadIdx.x] = mr;
    pq[threadIdx.x] = mq;
    const uint64_t first = (uint64_t)blockIdx.x * 256u;
    const uint32_t lim32 = (uint32_t)(n_rays - first < 256u ? n_rays - first : 256u);
    flush_structs<3>(sray, pr, reinterpret_cast<float4*>(rays) + first * 3u, 0u, first < n_rays ? lim32 : 0u);
    flush_structs<4>(spoi, pq, reinterpret_cast<float4*>(pois) + first * 4u, 0u, first < n_rays ? lim32 : 0u);
}

// code.cl:581-598
__global__ void __launch_bounds__(256) k_bouncePaths(const PoiAoS* pois, RayAoS* rays, int32_t* seeds, uint32_t total, uint32_t gsz) {
    __shared__ float4 sray[256 * 3];
    __shared__ uint8_t pr[256];
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lim = gsz < total ? gsz : total;
    uint8_t mr = 0;
    if (id < lim) {
        const PoiAoS* pp = &pois[id];
        if (pp->matId >= 0) {
            Poi poi;
            poi.p = mk3(pp->px, pp->py, pp->pz);
            poi.n = mk3(pp->nx, pp->ny, pp->nz);
            int32_t s = seeds[id];
            Ray r = bounce_ray(poi, s);
            seeds[id] = s;
            put_ray(sray, r);
            mr = kRayAll;
        } else {
            put_dead(sray);
            mr = kRayTail;
        }
    }
    pr[threadIdx.x] = mr;
    flush_structs<3>(sray, pr, rays, blockIdx.x * 256u, lim);
}

// code.cl:600-629
__global__ void __launch_bounds__(256) k_lightRender(PoiAoS* pois, RayAoS* rays, float4* acu, F16 light, uint32_t total, uint32_t gsz) {
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= gsz || id >= total) return;
    Ray ray = load_ray(&rays[id]);
    if (ray.mint == ray.maxt) return;
    f3 irr = norm3(ld3(light.v + 6));
    if (!light_visible(ray, ld3(light.v), ld3(light.v + 3), light.v[9])) return;
    store_dead(&rays[id]);
    pois[id].matId = -1;
    float4 a = acu[id];
    a.x += irr.x; a.y += irr.y; a.z += irr.z; a.w += 1.0f;
    acu[id] = a;
}

// code.cl:631-673
__global__ void __launch_bounds__(256) k_initShadowTrace(RayAoS* shadow, const PoiAoS* pois, uint32_t total, F16 light,
                                                          int32_t* seeds, uint32_t gsz) {
    __shared__ float4 sray[256 * 3];
    __shared__ uint8_t pr[256];
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lim = gsz < total ? gsz : total;
    uint8_t mr = 0;
    if (id < lim) {
        const PoiAoS* pp = &pois[id];
        if (pp->matId < 0) {
            put_dead(sray);
            mr = kRayTail;
        } else {
            Poi poi;
            poi.p = mk3(pp->px, pp->py, pp->pz);
            poi.n = mk3(pp->nx, pp->ny, pp->nz);
            int32_t s = seeds[id];
            Ray r = shadow_ray(poi, ld3(light.v), ld3(light.v + 3), ld3(light.v + 6), light.v[9], s);
            seeds[id] = s;
            put_ray(sray, r);
            mr = kRayAll;
        }
    }
    pr[threadIdx.x] = mr;
    flush_structs<3>(sray, pr, shadow, blockIdx.x * 256u, lim);
}

// code.cl:675-800 (spheres), 802-935 (triangles, per-primitive material), 937-1070 (mesh, one material): one template,
// three instantiations.  The traversal is pt_trace.hpp's (prepared triangles, wave-uniform loop for single-cell grids).
PT_DEV GridArgs mk_set(const float4* prims, const uint32_t* off, const Box8& b, uint32_t n, uint32_t exit_far) {
    GridArgs S;
    S.prims = prims; S.normals = nullptr; S.matid = nullptr; S.off = off;
    for (int i = 0; i < 8; ++i) S.bound[i] = b.v[i];
    S.n = n; S.mesh_matid = 0; S.kind = 0; S.fast_ok = 0; S.lds_off = kNoLds; S.exit_is_far_face = exit_far;
    for (int i = 0; i < 3; ++i) { S.delta[i] = 0.0f; S.rdelta[i] = 0.0f; }   // the optimistic kernel's (these kernels divide for themselves)
    S.walk_ok = 0; S.nslots = 0;
    return S;
}
template <int KIND>
__global__ void __launch_bounds__(256) k_closest(uint32_t total, PoiAoS* pois, RayAoS* rays, const float4* prims,
                                                  const float4* normals, const uint32_t* matid, uint32_t mesh_matid,
                                                  const uint32_t* off, Box8 bound8, uint32_t n, uint32_t exit_far, uint32_t gsz) {
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= gsz || id >= total) return;
    Ray ray = load_ray(&rays[id]);
    if (ray.mint == ray.maxt) return;
    BoxHit bh = inter_aabb(ray, mk_box(bound8));
    if (!bh.v) return;
    bool unused = false;
    Hit ch = trace_set<KIND, false>(ray, bh, mk_set(prims, off, bound8, n, exit_far), unused);
    if (ch.idx == UINT32_MAX) return;
    rays[id].maxt = ch.t;
    // p, normal, matId -- never atte (SURVEY 8a hazard 2); on a miss the previous vertex stays live (hazard 3)
    f3 p = fma3(ch.t, ray.d, ray.o);   // getPoint, code.cl:87
    f3 nrm;
    if (KIND == SPHERES) {
        nrm = norm3(sub3(p, ld3(prims[ch.idx])));
    } else {
        const float4* nn = normals + 3u * (size_t)ch.idx;
        float w = 1.0f - ch.beta - ch.gamma;                         // code.cl:409-411
        nrm = norm3(fma3(ch.gamma, ld3(nn[2]), fma3(w, ld3(nn[0]), scl3(ch.beta, ld3(nn[1])))));
    }
    PoiAoS* pp = &pois[id];
    float4* q = reinterpret_cast<float4*>(pp);
    q[0] = make_float4(p.x, p.y, p.z, 0.0f);
    q[1] = make_float4(nrm.x, nrm.y, nrm.z, 0.0f);
    pp->matId = (int32_t)(matid ? matid[ch.idx] : mesh_matid);
}

// code.cl:1073-1193, 1195-1321: blocked -> mint = maxt = t (the "dead ray" mark the next kernels and sceneRender test)
template <int KIND>
__global__ void __launch_bounds__(256) k_anyhit(uint32_t total, RayAoS* shadow, const float4* prims, const uint32_t* off,
                                                 Box8 bound8, uint32_t n, uint32_t exit_far, uint32_t gsz) {
    __shared__ float4 sray[256 * 3];
    __shared__ uint8_t pr[256];
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lim = gsz < total ? gsz : total;
    uint8_t mr = 0;
    if (id < lim) {
        Ray sh = load_ray(&shadow[id]);
        if (!(sh.mint == sh.maxt)) {
            BoxHit bh = inter_aabb(sh, mk_box(bound8));
            if (bh.v) {
                bool unused = false;
                Hit ch = trace_set<KIND, true>(sh, bh, mk_set(prims, off, bound8, n, exit_far), unused);
                // a free ray gets its own mint / maxt written back by the reference (code.cl:1186-1190): same bits, so nothing to store
                if (ch.idx != UINT32_MAX) {
                    sh.mint = ch.t;
                    sh.maxt = ch.t;
                    put_ray(sray, sh);
                    mr = kRayAll;
                }
            }
        }
    }
    pr[threadIdx.x] = mr;
    flush_structs<3>(sray, pr, shadow, blockIdx.x * 256u, lim);
}

// meshTrace / triangleTrace and triangleShadowTrace over a grid with n > 1: the same kernels, with the walk whose triangle tests the wave
// shares (pt_trace_coop.hpp).  Every lane stays in (a lane without a live ray walks nothing and tests for the others); 16 KB of dynamic
// LDS per block for the waves' exchange rows.  Same per-ray cells, primitives, windows and results as the per-lane walk above.
__global__ void __launch_bounds__(256) k_closestMesh(uint32_t total, PoiAoS* pois, RayAoS* rays, const float4* prims, const float4* normals,
                                                      const uint32_t* matid, uint32_t mesh_matid, const uint32_t* off, Box8 bound8, uint32_t n, uint32_t gsz) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = id < gsz && id < total;
    Ray ray = {};
    BoxHit bh = {};
    bool want = false;
    if (mine) {
        ray = load_ray(&rays[id]);
        if (!(ray.mint == ray.maxt)) {
            bh = inter_aabb(ray, mk_box(bound8));
            want = bh.v;
        }
    }
    bool unused = false;
    const Hit ch = trace_dda_coop<COOP_CLOSEST, false, false>(want, ray, RayRcp{0.0f, 0.0f, 0.0f}, bh, mk_set(prims, off, bound8, n, 0u), unused);
    if (!want || ch.idx == UINT32_MAX) return;
    rays[id].maxt = ch.t;
    const f3 p = fma3(ch.t, ray.d, ray.o);   // getPoint, code.cl:87
    const float4* nn = normals + 3u * (size_t)ch.idx;
    const float w = 1.0f - ch.beta - ch.gamma;                         // code.cl:409-411
    const f3 nrm = norm3(fma3(ch.gamma, ld3(nn[2]), fma3(w, ld3(nn[0]), scl3(ch.beta, ld3(nn[1])))));
    PoiAoS* pp = &pois[id];
    float4* q = reinterpret_cast<float4*>(pp);
    q[0] = make_float4(p.x, p.y, p.z, 0.0f);
    q[1] = make_float4(nrm.x, nrm.y, nrm.z, 0.0f);
    pp->matId = (int32_t)(matid ? matid[ch.idx] : mesh_matid);
}

__global__ void __launch_bounds__(256) k_anyhitMesh(uint32_t total, RayAoS* shadow, const float4* prims, const uint32_t* off, Box8 bound8, uint32_t n, uint32_t gsz) {
    __shared__ float4 sray[256 * 3];
    __shared__ uint8_t pr[256];
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lim = gsz < total ? gsz : total;
    Ray sh = {};
    BoxHit bh = {};
    bool want = false;
    if (id < lim) {
        sh = load_ray(&shadow[id]);
        if (!(sh.mint == sh.maxt)) {
            bh = inter_aabb(sh, mk_box(bound8));
            want = bh.v;
        }
    }
    bool unused = false;
    const Hit ch = trace_dda_coop<COOP_ANY_FIRST, false, false>(want, sh, RayRcp{0.0f, 0.0f, 0.0f}, bh, mk_set(prims, off, bound8, n, 0u), unused);
    uint8_t mr = 0;
    if (want && ch.idx != UINT32_MAX) {
        sh.mint = ch.t;
        sh.maxt = ch.t;
        put_ray(sray, sh);
        mr = kRayAll;
    }
    pr[threadIdx.x] = mr;
    flush_structs<3>(sray, pr, shadow, blockIdx.x * 256u, lim);
}

// code.cl:1323-1364.  `nmat` guards the material fetch: an out-of-range id (undefined
// behaviour in the reference) shades nothing here instead of reading foreign memory.
__global__ void __launch_bounds__(256) k_sceneRender(float4* acu, PoiAoS* pois, const RayAoS* shadow, const float4* material,
                                                      uint32_t nmat, F16 light, uint32_t total, uint32_t gsz) {
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= gsz || id >= total) return;
    PoiAoS* pp = &pois[id];
    int32_t m = pp->matId;
    if (m < 0 || (uint32_t)m >= nmat) return;
    Poi poi;
    poi.p = mk3(pp->px, pp->py, pp->pz);
    poi.n = mk3(pp->nx, pp->ny, pp->nz);
    poi.atte = mk3(pp->ax, pp->ay, pp->az);
    Ray sh = load_ray(&shadow[id]);
    float4 c4 = material[m];
    f3 c = shade_vertex(poi, sh, mk3(c4.x, c4.y, c4.z), ld3(light.v), ld3(light.v + 3), ld3(light.v + 6), light.v[9]);
    pp->ax = poi.atte.x; pp->ay = poi.atte.y; pp->az = poi.atte.z;
    float4 a = acu[id];
    a.x += c.x; a.y += c.y; a.z += c.z; a.w += 1.0f;
    acu[id] = a;
}

// code.cl:1366-1386; `radiance` (optional) receives the un-scaled sequential fp32 sums.
// One wave per 64 pixels.  The reference's work-item walks its own run of rpp float4s (4 KB apart
// from its neighbour's at rpp = 256: every lane on a different line, 4x over-fetch measured); here
// the wave stages 64 pixels x 16 samples through LDS with coalesced 256-B runs, then each lane adds
// its pixel's samples in the reference's order (the fp32 sum is order-dependent).
constexpr int kResS = 16;   // samples per staged chunk
__global__ void __launch_bounds__(64) k_copyToPixel(uchar4* pixel, const float4* acu, float m, uint32_t pixels, uint32_t rpp,
                                                     uint32_t gsz, float4* radiance) {
    __shared__ float4 tile[64][kResS + 1];   // +1: lanes 16 apart land on different banks for ds_read_b128
    const uint32_t lane = threadIdx.x;
    const uint32_t pix0 = blockIdx.x * 64u;
    const uint32_t id = pix0 + lane;
    const uint32_t lim = pixels < gsz ? pixels : gsz;
    float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t s0 = 0; s0 < rpp; s0 += kResS) {
        const uint32_t ns = (rpp - s0) < (uint32_t)kResS ? (rpp - s0) : (uint32_t)kResS;
#pragma unroll
        for (int it = 0; it < kResS; ++it) {
            const uint32_t idx = (uint32_t)it * 64u + lane;
            const uint32_t px = idx / kResS, s = idx % kResS;
            if (s < ns && pix0 + px < lim) tile[px][s] = acu[(size_t)(pix0 + px) * rpp + s0 + s];
        }
        __syncthreads();
        if (id < lim)
            for (uint32_t s = 0; s < ns; ++s) {
                const float4 v = tile[lane][s];
                c.x += v.x; c.y += v.y; c.z += v.z; c.w += v.w;
            }
        __syncthreads();
    }
    if (id >= lim) return;
    if (radiance) radiance[id] = c;
    float sc = 255.0f * m;
    c.x = cl_clamp((c.x * sc) * 1.8f, 0.0f, 255.0f);
    c.y = cl_clamp((c.y * sc) * 1.8f, 0.0f, 255.0f);
    c.z = cl_clamp((c.z * sc) * 1.8f, 0.0f, 255.0f);
    if (pixel) pixel[id] = make_uchar4((unsigned char)f2u(c.x), (unsigned char)f2u(c.y), (unsigned char)f2u(c.z), 255);
}

// rpp <= 4 (the page's default is 1: progressive passes): a pixel's samples span at most one 64-byte sector, so the work-item
// walk of the reference is already coalesced -- no staging, 256-thread blocks (k_copyToPixel at rpp 1: 67 us per 1080p frame,
// this: see profiles/README.md)
__global__ void __launch_bounds__(256) k_copyToPixelSmall(uchar4* pixel, const float4* acu, float m, uint32_t pixels, uint32_t rpp,
                                                           uint32_t gsz, float4* radiance) {
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lim = pixels < gsz ? pixels : gsz;
    if (id >= lim) return;
    float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4* a = acu + (size_t)id * rpp;
    for (uint32_t s = 0; s < rpp; ++s) {
        const float4 v = a[s];
        c.x += v.x; c.y += v.y; c.z += v.z; c.w += v.w;
    }
    if (radiance) radiance[id] = c;
    float sc = 255.0f * m;
    c.x = cl_clamp((c.x * sc) * 1.8f, 0.0f, 255.0f);
    c.y = cl_clamp((c.y * sc) * 1.8f, 0.0f, 255.0f);
    c.z = cl_clamp((c.z * sc) * 1.8f, 0.0f, 255.0f);
    if (pixel) pixel[id] = make_uchar4((unsigned char)f2u(c.x), (unsigned char)f2u(c.y), (unsigned char)f2u(c.z), 255);
}

// splitmix32-style seed fill: s[id] = 1 + (mix(id ^ 0x9E3779B9 ^ base) mod 2147483646), ids global
// (SURVEY 8d config 4); the reference seeds with Math.random() on the host (A10 code.js:1140-1146).
__global__ void __launch_bounds__(256) k_seedFill(int32_t* seeds, uint64_t first, uint64_t count, uint32_t base) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t x = (uint32_t)(first + i) ^ 0x9E3779B9u ^ base;
    uint32_t z = x + 0x9E3779B9u;
    z = (z ^ (z >> 16)) * 0x85EBCA6Bu;
    z = (z ^ (z >> 13)) * 0xC2B2AE35u;
    z = z ^ (z >> 16);
    seeds[i] = (int32_t)(1u + (z % 2147483646u));
}

// Diagnostic: evaluates one primitive of the numerics contract element-wise, so tests can
// compare the device's bits with the CPU's over millions of inputs (tests/test_gpu_numerics.py).
__global__ void __launch_bounds__(256) k_numerics(int op, const float* a, const float* b, float* out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool vec = op >= 20 && op != 25;
    float x = vec ? 0.0f : a[i], y = (b && !vec) ? b[i] : 0.0f, r = 0.0f, t;
    switch (op) {
        case 0: r = x / y; break;
        case 1: r = cl_sqrt(x); break;
        case 2: cl_sincos(x, r, t); break;
        case 3: cl_sincos(x, t, r); break;
        case 4: { int32_t s = __float_as_int(x); r = get_rand(s); break; }          // LCG step + float map
        case 5: { int32_t s = __float_as_int(x); r = __int_as_float(lcg_next(s)); break; }  // raw next state
        case 6: r = cl_min(x, y); break;
        case 7: r = cl_max(x, y); break;
        case 8: r = cl_fmin(x, y); break;
        case 9: r = cl_fmax(x, y); break;
        case 10: { f3 v = norm3(mk3(x, y, 1.0f)); r = v.x; break; }
        case 11: concentric(x, y, r, t); break;
        case 12: concentric(x, y, t, r); break;
        case 13: r = __int_as_float(f2i(x)); break;
        case 14: r = __uint_as_float(f2u(x)); break;
        case 25: r = cl_clamp(x, 0.0f, y); break;
        // float3 arguments: three consecutive floats per element
        case 20: r = dot3(ld3(a + 3 * i), ld3(b + 3 * i)); break;
        case 22: r = len3(ld3(a + 3 * i)); break;
        case 23: r = len3(sub3(ld3(a + 3 * i), ld3(b + 3 * i))); break;                    // distance
        case 26: r = cl_mad(a[3 * i], a[3 * i + 1], a[3 * i + 2]); break;
        case 21: { f3 v = cross3(ld3(a + 3 * i), ld3(b + 3 * i)); out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z; return; }
        case 24: { f3 v = norm3(ld3(a + 3 * i)); out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z; return; }
        case 27: {   // the four contraction shapes of the OpenCL front end: a*b+c, a*b-c, c-a*b, a*b+c*a
            const float p = a[3 * i], q = a[3 * i + 1], c = a[3 * i + 2];
            out[4 * i] = cl_fma(p, q, c); out[4 * i + 1] = cl_fma(p, q, -c); out[4 * i + 2] = cl_fma(-p, q, c); out[4 * i + 3] = cl_fma(p, q, c * p);
            return;
        }
        default: break;
    }
    out[i] = r;
}

// Diagnostic: counts, over `count` pseudo-random (n, d) pairs inside the guard window, how often the
// shared-reciprocal forms differ from the compiler's correctly rounded division.
//   out[0] div_shared(n,d,rcp_refined(d)) != n/d     out[1] rcp_refined(d) != 1/d
//   out[2] div_shared(1,d,r) != 1/d                  out[3] 3-op form (one refinement) != n/d
//   out[4..7] bit patterns (n, d) of the first mismatch of kind 0 / kind 3      out[12] div_exact3(n,d,r) != n/d (the form the kernels use)
// mode 0: random mantissas and exponents; mode 1: d sweeps EVERY mantissa (count = 2^23 * exponents), n random;
// mode 2: as 0 without the zero numerators
__global__ void __launch_bounds__(256) k_divCheck(int mode, uint64_t seed, uint64_t count, unsigned long long* out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0;
    if (mode == 3) {
        // every one of the 2^32 bit patterns as a denominator: rcp_refined(d) against 1.0f/d (NaN == NaN);
        // out[1] = mismatches, out[10]/out[11] = smallest / largest |d| bit pattern that mismatched
        unsigned long long lo = ~0ull, hi = 0;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
            const float d = __uint_as_float((uint32_t)i);
            const float a = rcp_refined(d), b = 1.0f / d;
            const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
            if (!same) { ++m1; const unsigned long long k = (uint32_t)i & 0x7FFFFFFFu; lo = k < lo ? k : lo; hi = k > hi ? k : hi; }
        }
        if (m1) { atomicAdd(&out[1], m1); atomicMin(&out[10], lo); atomicMax(&out[11], hi); }
        return;
    }
    if (mode == 5) {
        // every one of the 2^32 bit patterns: cl_sqrt (the 9-operation core + denormal fallback) against the compiler's correctly rounded
        // sqrt, and the bare core alone (out[2]: it may differ only inside (0, 2^-96), out[3] counts the others); out[1] = cl_sqrt mismatches
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
            const float x = __uint_as_float((uint32_t)i);
            const float want = __builtin_sqrtf(x), got = cl_sqrt(x), core = sqrt_core(x);
            if (!((__float_as_uint(got) == __float_as_uint(want)) || (got != got && want != want))) { ++m1; atomicMin(&out[10], (unsigned long long)i); atomicMax(&out[11], (unsigned long long)i); out[4] = __float_as_uint(got); out[5] = __float_as_uint(want); }
            if (!((__float_as_uint(core) == __float_as_uint(want)) || (core != core && want != want))) {
                ++m2;
                if (!(__builtin_fabsf(x) > 0.0f && __builtin_fabsf(x) < 1.2621774e-29f)) ++m3;   // a core mismatch with |x| OUTSIDE (0, 2^-96) would break the claim
            }
        }
        if (m1) atomicAdd(&out[1], m1);
        if (m2) atomicAdd(&out[2], m2);
        if (m3) atomicAdd(&out[3], m3);
        return;
    }
    if (mode == 4) {
        // EVERY mantissa pair: d = 1.dm, n = 1.nm, dm in [seed, seed + count), nm in [0, 2^23).  Powers of two scale every
        // step of div_exact3 exactly while nothing leaves the normal range (the windows guarantee that) and rcp_refined is
        // RN(1/d) for every d in its window (mode 3), so one binade pair stands for all of them; signs are symmetric.
        // out[3] = mismatches, out[6]/out[7] = first mismatching (n, d)
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (count << 23); i += stride) {
            const float d = __uint_as_float(0x3F800000u | (uint32_t)(seed + (i >> 23)));
            const float n = __uint_as_float(0x3F800000u | (uint32_t)(i & 0x7FFFFFu));
            if (__float_as_uint(div_exact3(n, d, rcp_refined(d))) != __float_as_uint(n / d)) {
                if (!m3 && !atomicAdd(&out[9], 1ull)) { out[6] = __float_as_uint(n); out[7] = __float_as_uint(d); }
                ++m3;
            }
        }
        if (m3) atomicAdd(&out[3], m3);
        return;
    }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        uint64_t h = (i + seed) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; h *= 0x94D049BB133111EBull; h ^= h >> 29;
        uint32_t dm, de;
        if (mode == 1) { dm = (uint32_t)(i & 0x7FFFFFu); de = 127u - 40u + (uint32_t)((i >> 23) % 81u); }
        else { dm = (uint32_t)(h & 0x7FFFFFu); de = 127u - 40u + (uint32_t)((h >> 23) % 81u); }
        const uint32_t ds = (uint32_t)(h >> 31) & 0x80000000u;
        const float d = __uint_as_float(ds | (de << 23) | dm);
        const uint32_t nm = (uint32_t)(h >> 32) & 0x7FFFFFu, ne = 127u - 60u + (uint32_t)((h >> 55) % 121u);
        const uint32_t ns = (uint32_t)(h >> 8) & 0x80000000u;
        float n = __uint_as_float(ns | (ne << 23) | nm);
        if (mode != 2 && (h & 0xFFF000000ull) == 0) n = __uint_as_float(ns);   // sprinkle +-0 numerators (not in mode 2)
        const float ref = n / d;
        const float r = rcp_refined(d);
        const float q = div_shared(n, d, r);
        const float inv = 1.0f / d;
        const float q1 = div_shared(1.0f, d, r);
        float q3 = n * r;
        q3 = __builtin_fmaf(__builtin_fmaf(-d, q3, n), r, q3);
        if (__float_as_uint(div_exact3(n, d, r)) != __float_as_uint(ref)) ++m4;
        if (__float_as_uint(q) != __float_as_uint(ref)) { if (!m0 && !atomicAdd(&out[8], 1ull)) { out[4] = __float_as_uint(n); out[5] = __float_as_uint(d); } ++m0; }
        if (__float_as_uint(r) != __float_as_uint(inv)) ++m1;
        if (__float_as_uint(q1) != __float_as_uint(inv)) ++m2;
        if (__float_as_uint(q3) != __float_as_uint(ref)) { if (!m3 && !atomicAdd(&out[9], 1ull)) { out[6] = __float_as_uint(n); out[7] = __float_as_uint(d); } ++m3; }
    }
    if (m0) atomicAdd(&out[0], m0);
    if (m1) atomicAdd(&out[1], m1);
    if (m2) atomicAdd(&out[2], m2);
    if (m3) atomicAdd(&out[3], m3);
    if (m4) atomicAdd(&out[12], m4);
}

}  // namespace pt

// ---- launchers (C++ linkage, called by the C-ABI layer) -----------------------------------
namespace pt {

static inline dim3 grid1(uint64_t n, unsigned b = 256) { return dim3((unsigned)((n + b - 1) / b)); }

void launch_sizeof(hipStream_t s, bool ray, uint32_t* out) {
    if (ray) hipLaunchKernelGGL(k_sizeofRay, dim3(1), dim3(64), 0, s, out);
    else hipLaunchKernelGGL(k_sizeofPoi, dim3(1), dim3(64), 0, s, out);
}
void launch_initAcu(hipStream_t s, void* acu, uint32_t total, uint32_t gsz) {
    if (!gsz) return;
    hipLaunchKernelGGL(k_initAcu, grid1(gsz), dim3(256), 0, s, (float4*)acu, total, gsz);
}
void launch_lensDraws(hipStream_t s, void* seeds, void* uv, uint32_t cols, uint32_t rows, uint32_t gx, uint32_t gy,
                      uint32_t row0, uint32_t nrows) {
    if (!cols) return;
    hipLaunchKernelGGL(k_lensDraws, grid1(cols, 64), dim3(64), 0, s, (int32_t*)seeds, (float2*)uv, cols, rows, gx, gy, row0, nrows);
}
void launch_initTrace(hipStream_t s, void* rays, void* pois, const void* uv, const float* bound, const float* cam,
                      float focal, float lens_rad, uint32_t rpp, uint32_t gx, uint32_t gy) {
    if (!gx || !gy || !rpp) return;
    Box8 b; F16 c;
    for (int i = 0; i < 8; ++i) b.v[i] = bound[i];
    for (int i = 0; i < 16; ++i) c.v[i] = cam[i];
    auto f2u_h = [](float f) -> uint32_t { return !(f == f) || f <= 0.0f ? 0u : f >= 4294967296.0f ? UINT32_MAX : (uint32_t)f; };   // = f2u() on the device
    const uint32_t cols = f2u_h(cam[14]), rows = f2u_h(cam[15]);
    const uint32_t wc = gx < cols ? gx : cols, wr = gy < rows ? gy : rows;
    const uint64_t n_rays = (uint64_t)cols * wr * rpp;
    if (!wc || !n_rays) return;
    hipLaunchKernelGGL(k_initTrace, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, s, (RayAoS*)rays, (PoiAoS*)pois, (const float2*)uv, b, c, focal,
                       lens_rad, rpp, wc, wr);
}
void launch_bouncePaths(hipStream_t s, const void* pois, void* rays, void* seeds, uint32_t total, uint32_t gsz) {
    if (!gsz) return;
    hipLaunchKernelGGL(k_bouncePaths, grid1(gsz), dim3(256), 0, s, (const PoiAoS*)pois, (RayAoS*)rays, (int32_t*)seeds, total, gsz);
}
static F16 mk16(const float* f) { F16 r; for (int i = 0; i < 16; ++i) r.v[i] = f[i]; return r; }
static Box8 mk8(const float* f) { Box8 r; for (int i = 0; i < 8; ++i) r.v[i] = f[i]; return r; }

void launch_lightRender(hipStream_t s, void* pois, void* rays, void* acu, const float* light, uint32_t total, uint32_t gsz) {
    if (!gsz) return;
    hipLaunchKernelGGL(k_lightRender, grid1(gsz), dim3(256), 0, s, (PoiAoS*)pois, (RayAoS*)rays, (float4*)acu, mk16(light), total, gsz);
}
void launch_initShadowTrace(hipStream_t s, void* shadow, const void* pois, uint32_t total, const float* light, void* seeds, uint32_t gsz) {
    if (!gsz) return;
    hipLaunchKernelGGL(k_initShadowTrace, grid1(gsz), dim3(256), 0, s, (RayAoS*)shadow, (const PoiAoS*)pois, total, mk16(light), (int32_t*)seeds, gsz);
}
void launch_closest(hipStream_t s, int kind, uint32_t total, void* pois, void* rays, const void* prims, const void* normals,
                    const void* matid, uint32_t mesh_matid, const void* off, const float* bound, uint32_t n, uint32_t exit_far, uint32_t gsz) {
    if (!gsz) return;
    if (kind == SPHERES)
        hipLaunchKernelGGL(k_closest<SPHERES>, grid1(gsz), dim3(256), 0, s, total, (PoiAoS*)pois, (RayAoS*)rays, (const float4*)prims,
                           (const float4*)normals, (const uint32_t*)matid, mesh_matid, (const uint32_t*)off, mk8(bound), n, exit_far, gsz);
    else if (n > 1u)
        hipLaunchKernelGGL(k_closestMesh, grid1(gsz), dim3(256), (size_t)kCoopWordsPerBlock * 4u, s, total, (PoiAoS*)pois, (RayAoS*)rays, (const float4*)prims,
                           (const float4*)normals, (const uint32_t*)matid, mesh_matid, (const uint32_t*)off, mk8(bound), n, gsz);
    else
        hipLaunchKernelGGL(k_closest<TRIANGLES>, grid1(gsz), dim3(256), 0, s, total, (PoiAoS*)pois, (RayAoS*)rays, (const float4*)prims,
                           (const float4*)normals, (const uint32_t*)matid, mesh_matid, (const uint32_t*)off, mk8(bound), n, exit_far, gsz);
}
void launch_anyhit(hipStream_t s, int kind, uint32_t total, void* shadow, const void* prims, const void* off, const float* bound,
                   uint32_t n, uint32_t exit_far, uint32_t gsz) {
    if (!gsz) return;
    if (kind == SPHERES)
        hipLaunchKernelGGL(k_anyhit<SPHERES>, grid1(gsz), dim3(256), 0, s, total, (RayAoS*)shadow, (const float4*)prims, (const uint32_t*)off, mk8(bound), n, exit_far, gsz);
    else if (n > 1u)
        hipLaunchKernelGGL(k_anyhitMesh, grid1(gsz), dim3(256), (size_t)kCoopWordsPerBlock * 4u, s, total, (RayAoS*)shadow, (const float4*)prims, (const uint32_t*)off, mk8(bound), n, gsz);
    else
        hipLaunchKernelGGL(k_anyhit<TRIANGLES>, grid1(gsz), dim3(256), 0, s, total, (RayAoS*)shadow, (const float4*)prims, (const uint32_t*)off, mk8(bound), n, exit_far, gsz);
}
void launch_sceneRender(hipStream_t s, void* acu, void* pois, const void* shadow, const void* material, uint32_t nmat,
                        const float* light, uint32_t total, uint32_t gsz) {
    if (!gsz) return;
    hipLaunchKernelGGL(k_sceneRender, grid1(gsz), dim3(256), 0, s, (float4*)acu, (PoiAoS*)pois, (const RayAoS*)shadow, (const float4*)material,
                       nmat, mk16(light), total, gsz);
}
void launch_copyToPixel(hipStream_t s, void* pixel, const void* acu, float m, uint32_t pixels, uint32_t rpp, uint32_t gsz, void* radiance) {
    if (!gsz) return;
    if (rpp <= 4u) {
        hipLaunchKernelGGL(k_copyToPixelSmall, grid1(gsz), dim3(256), 0, s, (uchar4*)pixel, (const float4*)acu, m, pixels, rpp, gsz, (float4*)radiance);
        return;
    }
    hipLaunchKernelGGL(k_copyToPixel, grid1(gsz, 64), dim3(64), 0, s, (uchar4*)pixel, (const float4*)acu, m, pixels, rpp, gsz, (float4*)radiance);
}
void launch_numerics(hipStream_t s, int op, const void* a, const void* b, void* out, uint64_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_numerics, grid1(n), dim3(256), 0, s, op, (const float*)a, (const float*)b, (float*)out, n);
}
void launch_divCheck(hipStream_t s, int mode, uint64_t seed, uint64_t count, void* out16) {
    hipLaunchKernelGGL(k_divCheck, dim3(256 * 32), dim3(256), 0, s, mode, seed, count, (unsigned long long*)out16);
}
void launch_seedFill(hipStream_t s, void* seeds, uint64_t first, uint64_t count, uint32_t base) {
    if (!count) return;
    hipLaunchKernelGGL(k_seedFill, grid1(count), dim3(256), 0, s, (int32_t*)seeds, first, count, base);
}

}  // namespace pt
