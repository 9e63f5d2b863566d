// pt_kernels_fused_sm.hip -- the fused pass for scenes WITH uniform grids (meshes, n_slabs > 1): one launch per progressive
// pass like k_fusedPass (pt_kernels_fused.hip), same per-ray arithmetic in the same per-ray order, different SCHEDULE.
//
// Why: in k_fusedPass<*, true> the 64 lanes of a wave move through the path in lock step -- every grid walk (per-lane 3-axis DDA,
// A10 code.cl:694-786 and its four copies) costs the wave its LONGEST walk, and a lane whose ray misses a mesh's box idles for all
// of it: 24 % lane utilisation on cornell_teapot3 (profiles/r1l_final).  A ray's result does not depend on when its operations
// run, only on their order within the ray.  So here every lane carries its own position in the path (stage, segment, light, the
// grid sets still to walk) and the wave repeatedly picks ONE stage and runs it for the lanes that are at it:
//
//   BEGIN   bounce / primary ray, closest hit over the single-cell sets (wave-uniform loops, scalar loads)      -> WALK | POST
//   WALK    one job = one grid set: box test + axis set-up, then the re-phased DDA (close / open cells, one primitive per lane);
//           closest-hit and any-hit (shadow) jobs of different lanes, different sets, different segments share the loop
//   POST    vertex (p, normal, matId) from the segment's champion; lightRender on the primary segment           -> SHADOW
//   SHADOW  light l: shadow ray (2 RNG draws), any-hit over the single-cell sets                               -> WALK | SHADE
//   SHADE   light l: sceneRender; next light, next segment or the end                                        -> SHADOW | BEGIN | DONE
//
// The stage with the most lanes waiting runs next (POST and SHADE are cheap and run whenever anybody waits).  A lane that missed the
// teapot's box is three stages ahead while its neighbour still walks the teapot; nobody waits for the slowest walk of each step,
// only -- at the very end -- for the slowest PATH, and with a fixed number of segments per path (no Russian roulette in the
// reference) path lengths differ far less than walk lengths.
//
// Per-lane order of operations == the reference's kernel sequence for that ray id (A10 code.js:1806-1854), with two re-orderings
// that cannot change any value:
//   * the vertex of a segment is computed once from the final champion instead of after every set that hits (each hit overwrites
//     p, normal, matId completely; atte is not touched, SURVEY 8a hazard 2);
//   * all single-cell sets are walked before the grid sets.  launch_fused only picks this kernel when that IS the upload order
//     (every n == 1 set before every n > 1 set: loose spheres and triangles always come first, A10 code.js:1809-1813), so it is
//     not a re-ordering at all.
// Only the optimistic arithmetic (FAST, pt_trace.hpp) is built: a sample whose rays leave the guard windows sets its bit in the
// defer mask and is redone by k_fusedPass<false, true>, as before.  Requirements checked by launch_fused_sm_ok(): every grid set
// holds triangles, all cell-offset tables fit the LDS budget, uniform sets precede grid sets; otherwise k_fusedPass runs.
#include "pt_trace.hpp"

#ifndef PT_SM_WAVES
#define PT_SM_WAVES 6
#endif
#ifndef PT_SM_BURST
#define PT_SM_BURST 6      // primitive tests per visit of the WALK stage before the scheduler looks again
#endif
#ifndef PT_SM_TAU
#define PT_SM_TAU 12       // the walk stage yields to a waiting wave-uniform stage once this many lanes or fewer still walk
#endif

namespace pt {

namespace {

enum : uint32_t { S_BEGIN = 0u, S_WALK = 1u, S_POST = 2u, S_SHADOW = 3u, S_SHADE = 4u, S_DONE = 5u };
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kSetWords = 16;   // per-set record in LDS, for the stages where the set differs from lane to lane
// record: [0..2] lo, [3..5] hi, [6] n, [7] lds_off, [8,9] prims, [10,11] normals, [12,13] matid, [14] mesh_matid, [15] kind

PT_DEV uint32_t popc64(uint64_t m) { return (uint32_t)__builtin_popcountll(m); }
PT_DEV uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

}  // namespace

__global__ void __launch_bounds__(256, PT_SM_WAVES) k_fusedPassSM(const FusedArgs A, uint32_t* defer_mask) {
    __shared__ uint32_t s_tables[kLdsOffWords];
    __shared__ uint32_t s_sets[2 + kMaxMeshes][kSetWords];
    __shared__ float park_mem[7][256];   // accumulator (4) + attenuation (3) per lane, [word][lane]
    for (uint32_t s = 0; s < A.n_sets; ++s) {
        const GridArgs& S = A.sets[s];
        if (threadIdx.x == 0) {
            uint32_t* r = s_sets[s];
            for (int k = 0; k < 3; ++k) { r[k] = __float_as_uint(S.bound[k]); r[3 + k] = __float_as_uint(S.bound[4 + k]); }
            r[6] = S.n; r[7] = S.lds_off;
            r[8] = (uint32_t)(uintptr_t)S.prims; r[9] = (uint32_t)((uintptr_t)S.prims >> 32);
            r[10] = (uint32_t)(uintptr_t)S.normals; r[11] = (uint32_t)((uintptr_t)S.normals >> 32);
            r[12] = (uint32_t)(uintptr_t)S.matid; r[13] = (uint32_t)((uintptr_t)S.matid >> 32);
            r[14] = S.mesh_matid; r[15] = S.kind;
        }
        if (S.n == 1u || S.lds_off == kNoLds) continue;
        const uint32_t words = S.n * S.n * S.n + 1u;
        const uint32_t* src = (const uint32_t*)S.off;
        for (uint32_t k = threadIdx.x; k < words; k += 256u) s_tables[S.lds_off + k] = src[k];
    }
    __syncthreads();
    uint32_t grid_mask = 0u;
    for (uint32_t s = 0; s < A.n_sets; ++s) if (A.sets[s].n != 1u) grid_mask |= 1u << s;

    const uint64_t n_local = (uint64_t)A.nrows * A.width * A.rpp;
    const uint64_t lid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    float* const park = &park_mem[0][threadIdx.x];
    auto pput = [&](int w, float v) { park[w * 256] = v; };
    auto pget = [&](int w) { return park[w * 256]; };

    // ---- per-lane path state
    uint32_t st = S_DONE, seg = 0u, light = 0u, jobs = 0u;
    bool defer = false, wany = false, active = false;
    int32_t seed = 0;
    Ray ray;
    ray.o = mk3(0.f, 0.f, 0.f); ray.d = mk3(0.f, 0.f, 0.f); ray.mint = PT_INF; ray.maxt = PT_INF;
    Poi poi;
    poi.p = mk3(0.f, 0.f, 0.f); poi.n = mk3(0.f, 0.f, 0.f); poi.atte = mk3(1.f, 1.f, 1.f); poi.matId = -1;
    Hit ch;   // champion of the current ray over the sets walked so far
    ch.idx = kNone; ch.t = PT_INF; ch.beta = 0.f; ch.gamma = 0.f;
    uint32_t ch_set = 0u;
    // ---- per-lane walk state (one job = one grid set)
    uint32_t cur = 0u, wn = 1u, wtab = 0u, wi = 0u, wend = 0u, slabs = 0u;   // slabs: x | y << 10 | z << 20
    const float4* wprims = nullptr;
    float tnx = 0.f, tny = 0.f, tnz = 0.f, dtx = 0.f, dty = 0.f, dtz = 0.f, wt = 0.f, wcmax = 0.f, wtmax = 0.f;
    bool job_hit = false;

    if (lid < n_local) {
        st = S_BEGIN;
        const uint64_t lpix = lid / A.rpp;
        const uint32_t smp = (uint32_t)(lid - lpix * A.rpp);
        const uint32_t lrow = (uint32_t)(lpix / A.width);
        const uint32_t col = (uint32_t)(lpix - (uint64_t)lrow * A.width);
        const uint32_t row = A.row0 + lrow;
        Cam cam;
        cam.eye = ld3(A.cam); cam.U = ld3(A.cam + 3); cam.V = ld3(A.cam + 6); cam.W = ld3(A.cam + 9);
        cam.width = A.cam[12]; cam.height = A.cam[13];
        cam.cols = f2u(A.cam[14]); cam.rows = f2u(A.cam[15]);
        Box bound;
        bound.lo = mk3(A.bound[0], A.bound[1], A.bound[2]);
        bound.hi = mk3(A.bound[4], A.bound[5], A.bound[6]);
        seed = A.seeds[lid];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);   // initAcu when the pass is a frame's first (A10 code.cl:448-456)
        if (!A.fresh) acc = ((const float4*)A.acu)[lid];
        pput(0, acc.x); pput(1, acc.y); pput(2, acc.z); pput(3, acc.w);
        pput(4, 1.0f); pput(5, 1.0f); pput(6, 1.0f);
        // initTrace (code.cl:458-543) for this one ray
        f3 fp = focal_point(cam, (float)col, (float)row, A.focal_length);
        float cx, cy;
        if (A.rpp > 1) {
            const uint32_t side = f2u(cl_sqrt((float)A.rpp));
            const float delta = 1.0f / (float)side;
            const uint32_t i = smp / side, j = smp - i * side;
            cy = delta / 2.0f;
            for (uint32_t k = 0; k < i; ++k) cy += delta;
            cx = delta / 2.0f;
            for (uint32_t k = 0; k < j; ++k) cx += delta;
        } else {
            float2 c = ((const float2*)A.uv)[lpix];
            cx = c.x;
            cy = c.y;
        }
        ray = thin_lens_ray(cam, fp, A.lens_rad, cx, cy);
        clip_to(ray, bound);
    }

    const float4* material = (const float4*)A.material;
    for (;;) {
        const uint64_t m_begin = ballot(st == S_BEGIN), m_walk = ballot(st == S_WALK), m_post = ballot(st == S_POST);
        const uint64_t m_shadow = ballot(st == S_SHADOW), m_shade = ballot(st == S_SHADE);
        if ((m_begin | m_walk | m_post | m_shadow | m_shade) == 0ull) break;

        // ================= POST: the segment's vertex from its champion; lightRender on the primary segment ==================
        if (m_post) {
            if (st == S_POST) {
                if (ch.idx != kNone) {
                    const uint32_t* r = s_sets[ch_set];
                    poi.p = fma3(ch.t, ray.d, ray.o);   // getPoint, code.cl:87
                    const uint32_t* mid = (const uint32_t*)(((uint64_t)r[13] << 32) | r[12]);
                    if (r[15] == (uint32_t)KIND_SPHERES) {
                        const float4* sp = (const float4*)(((uint64_t)r[9] << 32) | r[8]);
                        poi.n = norm3(sub3(poi.p, ld3(sp[ch.idx])));
                    } else {
                        const float4* nn = (const float4*)(((uint64_t)r[11] << 32) | r[10]) + 3u * (size_t)ch.idx;
                        const float w = 1.0f - ch.beta - ch.gamma;  // code.cl:409-411
                        poi.n = norm3(fma3(ch.gamma, ld3(nn[2]), fma3(w, ld3(nn[0]), scl3(ch.beta, ld3(nn[1])))));
                    }
                    poi.matId = (int32_t)(mid ? mid[ch.idx] : r[14]);
                }
                if (seg == 0u) {
                    for (uint32_t l = 0; l < A.n_lights; ++l) {  // lightRender (code.cl:600-629), primary segment only
                        if (ray.mint == ray.maxt) continue;
                        const LightArgs& L = A.lights[l];
                        f3 irr = norm3(ld3(L.light + 6));
                        if (!light_visible(ray, ld3(L.light), ld3(L.light + 3), L.light[9])) continue;
                        ray.mint = PT_INF;
                        ray.maxt = PT_INF;
                        poi.matId = -1;
                        pput(0, pget(0) + irr.x); pput(1, pget(1) + irr.y); pput(2, pget(2) + irr.z); pput(3, pget(3) + 1.0f);
                    }
                }
                light = 0u;
                st = A.n_lights ? S_SHADOW : S_SHADE;
            }
        }

        // ================= SHADE: sceneRender for the light the first waiting lane is at ========================================
        {
            const uint64_t m = ballot(st == S_SHADE);
            if (m) {
                const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)light, (int)__builtin_ctzll(m));
                if (st == S_SHADE && light == l) {
                    if (l < A.n_lights) {
                        const LightArgs& L = A.lights[l];
                        if (poi.matId >= 0 && (uint32_t)poi.matId < A.nmat) {   // out-of-range id: shade nothing (see k_sceneRender)
                            const float4 c4 = material[poi.matId];
                            poi.atte = mk3(pget(4), pget(5), pget(6));
                            const f3 c = shade_vertex(poi, ray, mk3(c4.x, c4.y, c4.z), ld3(L.scene), ld3(L.scene + 3), ld3(L.scene + 6), L.scene[9]);
                            pput(4, poi.atte.x); pput(5, poi.atte.y); pput(6, poi.atte.z);
                            pput(0, pget(0) + c.x); pput(1, pget(1) + c.y); pput(2, pget(2) + c.z); pput(3, pget(3) + 1.0f);
                        }
                        ++light;
                    }
                    if (light < A.n_lights) st = S_SHADOW;
                    else if (seg < A.bounces) { ++seg; st = S_BEGIN; }
                    else {
                        st = S_DONE;
                        if (defer) atomicOr(&defer_mask[lid >> 5], 1u << (lid & 31u));   // the exact kernel redoes it from its untouched inputs
                        else {
                            A.seeds[lid] = seed;
                            ((float4*)A.acu)[lid] = make_float4(pget(0), pget(1), pget(2), pget(3));
                        }
                    }
                }
            }
        }

        // ================= scheduler ==========================================================================================
        // BEGIN and SHADOW are wave-uniform loops over the single-cell sets: a visit costs the same however few lanes are at it, and a
        // lane waiting there costs nothing.  WALK is per lane: a round costs the same however MANY lanes walk.  So: walk while more than
        // PT_SM_TAU lanes do; once only the tail of the slowest walks is left, let the lanes that are waiting run their stage (the tail
        // keeps its state and rejoins the loop with the next phase's walkers; it reaches its own stage one visit later, together with
        // whoever is there by then -- the stages are the same code for every segment).
        const uint64_t w_begin = ballot(st == S_BEGIN), w_walk = ballot(st == S_WALK), w_shadow = ballot(st == S_SHADOW);
        const uint32_t nb = popc64(w_begin), nw = popc64(w_walk);
        uint32_t ns = 0u, sl = 0u;
        if (w_shadow) {
            sl = (uint32_t)__builtin_amdgcn_readlane((int)light, (int)__builtin_ctzll(w_shadow));
            ns = popc64(ballot(st == S_SHADOW && light == sl));
        }
        if (nb == 0u && nw == 0u && ns == 0u) continue;
        const bool walk_now = nw > (uint32_t)PT_SM_TAU || (nb == 0u && ns == 0u);

        if (!walk_now && nb >= ns) {
            // ============= BEGIN: bounce ray (segments 1..), closest hit over the single-cell sets ================================
            if (st == S_BEGIN) {
                if (seg > 0u) {
                    if (poi.matId >= 0) ray = bounce_ray(poi, seed);
                    else { ray.mint = PT_INF; ray.maxt = PT_INF; }
                }
                if (!(ray.mint == ray.maxt)) defer = defer || !ray_guard(ray);   // a dead ray divides nothing
                ch.idx = kNone;
                for (uint32_t s = 0; s < A.n_sets; ++s) {
                    const GridArgs& S = A.sets[s];
                    if (S.n != 1u) continue;
                    if (ray.mint == ray.maxt) continue;
                    Box b;
                    b.lo = mk3(S.bound[0], S.bound[1], S.bound[2]); b.hi = mk3(S.bound[4], S.bound[5], S.bound[6]);
                    const BoxHit bh = inter_aabb_t<true, false>(ray, b);
                    if (!bh.v) continue;
                    const Hit h = (S.kind == KIND_SPHERES) ? trace_cell1<SPHERES, false, TRI_A10, true>(ray, bh, S) : trace_cell1<TRIANGLES, false, TRI_A10, true>(ray, bh, S);
                    if (h.idx == kNone) continue;
                    ch = h;
                    ch_set = s;
                    ray.maxt = h.t;
                }
                jobs = (ray.mint == ray.maxt) ? 0u : grid_mask;
                wany = false;
                active = false;
                st = jobs ? S_WALK : S_POST;
            }
        } else if (!walk_now) {
            // ============= SHADOW: shadow ray of light l and any-hit over the single-cell sets =====================================
            const uint32_t l = sl;
            if (st == S_SHADOW && light == l) {
                const LightArgs& L = A.lights[l];
                const bool path = poi.matId >= 0;  // initShadowTrace: a dead path draws nothing (code.cl:645-650)
                ray.o = mk3(0.f, 0.f, 0.f); ray.d = mk3(0.f, 0.f, 0.f); ray.mint = PT_INF; ray.maxt = PT_INF;
                if (path) {
                    ray = shadow_ray(poi, ld3(L.shadow), ld3(L.shadow + 3), ld3(L.shadow + 6), L.shadow[9], seed);
                    defer = defer || !ray_guard(ray);
                }
                for (uint32_t s = 0; s < A.n_sets; ++s) {
                    const GridArgs& S = A.sets[s];
                    if (S.n != 1u) continue;
                    if (!path || ray.mint == ray.maxt) continue;
                    Box b;
                    b.lo = mk3(S.bound[0], S.bound[1], S.bound[2]); b.hi = mk3(S.bound[4], S.bound[5], S.bound[6]);
                    const BoxHit bh = inter_aabb_t<true, false>(ray, b);
                    if (!bh.v) continue;
                    const Hit h = (S.kind == KIND_SPHERES) ? trace_cell1<SPHERES, true, TRI_A10, true>(ray, bh, S) : trace_cell1<TRIANGLES, true, TRI_A10, true>(ray, bh, S);
                    ray.maxt = h.t;
                    if (h.idx != kNone) ray.mint = h.t;
                }
                jobs = (path && !(ray.mint == ray.maxt)) ? grid_mask : 0u;
                wany = true;
                active = false;
                st = jobs ? S_WALK : S_SHADE;
            }
        } else {
            // ============= WALK: grid jobs.  Lanes without a running job open their next one; then PT_SM_BURST rounds of
            // "close / open cells until every walking lane holds a primitive, test one primitive per lane" ======================
            for (int round = 0; round < PT_SM_BURST; ++round) {
                // ---- open jobs: box test (interAABB at the head of every trace kernel) and axis set-up, per lane from the LDS records
                while (ballot(st == S_WALK && !active && jobs != 0u)) {
                    if (st == S_WALK && !active && jobs != 0u) {
                        cur = (uint32_t)__builtin_ctz(jobs);
                        jobs &= jobs - 1u;
                        const uint32_t* r = s_sets[cur];
                        Box b;
                        b.lo = mk3(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]));
                        b.hi = mk3(__uint_as_float(r[3]), __uint_as_float(r[4]), __uint_as_float(r[5]));
                        const BoxHit bh = inter_aabb_t<true, true>(ray, b);
                        if (bh.v) {
                            wn = r[6];
                            wtab = r[7];
                            wprims = (const float4*)(((uint64_t)r[9] << 32) | r[8]);
                            const Axis ax = axis_setup_t<true>(ray.o.x, ray.d.x, bh.tmin, b.lo.x, b.hi.x, wn, defer);
                            const Axis ay = axis_setup_t<true>(ray.o.y, ray.d.y, bh.tmin, b.lo.y, b.hi.y, wn, defer);
                            const Axis az = axis_setup_t<true>(ray.o.z, ray.d.z, bh.tmin, b.lo.z, b.hi.z, wn, defer);
                            tnx = ax.tnext; tny = ay.tnext; tnz = az.tnext;
                            dtx = ax.dt; dty = ay.dt; dtz = az.dt;
                            slabs = (uint32_t)ax.slab | ((uint32_t)ay.slab << 10) | ((uint32_t)az.slab << 20);
                            wt = bh.tmin;
                            wtmax = bh.tmax;
                            wcmax = __builtin_fminf(__builtin_fminf(tnx, tny), tnz);
                            const uint32_t cell = __umul24((slabs >> 20) & 1023u, wn * wn) + __umul24((slabs >> 10) & 1023u, wn) + (slabs & 1023u);
                            wi = s_tables[wtab + cell];
                            wend = s_tables[wtab + cell + 1u];
                            job_hit = false;
                            if (!wany) ch.t = ray.maxt;   // closest: champ_t starts at rays[id].maxt (code.cl:736); any-hit: likewise (:1131)
                            else ch.t = ray.maxt;
                            active = true;
                        } else if (jobs == 0u) {
                            st = wany ? S_SHADE : S_POST;   // the last set's box was missed: nothing more to walk
                        }
                    }
                }
                if (!ballot(st == S_WALK && active)) break;
                // ---- phase A: lanes whose cell list is exhausted close the cell and open the next one
                bool alive = st == S_WALK && active;
                while (ballot(alive && wi == wend)) {
                    if (alive && wi == wend) {
                        // a hit inside the cell ends the walk (code.cl:768-771); else step the axis whose plane was reached
                        if (job_hit) alive = false;
                        else {
                            const float t = wcmax;
                            bool out;
                            if (t == tnx) {
                                tnx += dtx;
                                const uint32_t sx = slabs & 1023u;
                                const bool fwd = ray.d.x >= 0;
                                out = t >= wtmax || (fwd ? sx + 1u == wn : sx == 0u);
                                slabs = fwd ? slabs + 1u : slabs - 1u;
                            } else if (t == tny) {
                                tny += dty;
                                const uint32_t sy = (slabs >> 10) & 1023u;
                                const bool fwd = ray.d.y >= 0;
                                out = t >= wtmax || (fwd ? sy + 1u == wn : sy == 0u);
                                slabs = fwd ? slabs + (1u << 10) : slabs - (1u << 10);
                            } else {
                                tnz += dtz;
                                const uint32_t sz = (slabs >> 20) & 1023u;
                                const bool fwd = ray.d.z >= 0;
                                out = t >= wtmax || (fwd ? sz + 1u == wn : sz == 0u);
                                slabs = fwd ? slabs + (1u << 20) : slabs - (1u << 20);
                            }
                            if (out) alive = false;
                            else {
                                wt = t;
                                wcmax = __builtin_fminf(__builtin_fminf(tnx, tny), tnz);
                                const uint32_t cell = __umul24((slabs >> 20) & 1023u, wn * wn) + __umul24((slabs >> 10) & 1023u, wn) + (slabs & 1023u);
                                wi = s_tables[wtab + cell];
                                wend = s_tables[wtab + cell + 1u];
                            }
                        }
                    }
                }
                // ---- phase B: one primitive per walking lane
                if (alive) {
                    const float4* __restrict__ p = wprims + 3u * (size_t)wi;
                    float ti, bb = 0.f, gm = 0.f;
                    const bool hit = tri_test<TRI_A10, true>(ray.o, ray.d, wt, wcmax, p[0], p[1], p[2], ti, bb, gm);
                    const bool better = (int)hit & (int)(ti < ch.t);
                    if (better) {
                        ch.t = ti;
                        job_hit = true;
                        if (!wany) { ch.idx = wi; ch.beta = bb; ch.gamma = gm; ch_set = cur; }
                    }
                    ++wi;
                    if (wany && better) alive = false;   // any-hit: the first accepted primitive ends the job (code.cl:1166-1171)
                }
                // ---- a finished job: fold its result into the ray, pick the next job or leave the stage
                if (st == S_WALK && active && !alive) {
                    active = false;
                    if (wany) {
                        ray.maxt = ch.t;                       // unchanged value re-stored when free (code.cl:1189-1192)
                        if (job_hit) { ray.mint = ch.t; jobs = 0u; }   // blocked: the ray is dead for the remaining sets
                    } else if (job_hit) ray.maxt = ch.t;
                    if (jobs == 0u) st = wany ? S_SHADE : S_POST;
                }
            }
        }
    }
}

bool launch_fused_sm_ok(const FusedArgs& a) {
    bool any_grid = false, seen_grid = false;
    uint64_t words = 0;
    for (uint32_t i = 0; i < a.n_sets; ++i) {
        const GridArgs& S = a.sets[i];
        if (S.n == 1u) { if (seen_grid) return false; continue; }
        seen_grid = any_grid = true;
        if (S.kind != (uint32_t)KIND_TRIANGLES || S.n > 1023u || !S.fast_ok) return false;
        words += (uint64_t)S.n * S.n * S.n + 1u;
    }
    return any_grid && words <= kLdsOffWords;
}

void launch_fused_sm(hipStream_t s, const FusedArgs& a, uint32_t* defer_mask) {
    const uint64_t n = (uint64_t)a.nrows * a.width * a.rpp;
    if (!n) return;
    FusedArgs b = a;
    uint32_t used = 0;
    for (uint32_t i = 0; i < b.n_sets; ++i) {
        b.sets[i].lds_off = kNoLds;
        if (b.sets[i].n > 1u) { b.sets[i].lds_off = used; used += b.sets[i].n * b.sets[i].n * b.sets[i].n + 1u; }
    }
    hipLaunchKernelGGL(k_fusedPassSM, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, b, defer_mask);
}

}  // namespace pt
