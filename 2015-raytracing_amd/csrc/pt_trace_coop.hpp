// pt_trace_coop.hpp -- the grid walk of the fused pass with the triangle tests SHARED by the wave.
//
// trace_dda (pt_trace.hpp) lets every lane test one triangle of its own cell per trip.  On an incoherent wave (bounce and shadow rays
// against a mesh in an n^3 grid) a quarter of the lanes hold a triangle at any time, and a 60-instruction test issues for the whole
// wave however few do: the walk is VALU-issue-bound at 24 % lane utilisation (cornell_teapot3).
//
// Here the walk alternates two wave-wide phases:
//   A  every live lane steps its own DDA through EMPTY cells until it stands in a cell that holds primitives (or has left the grid);
//   B  the (ray, primitive) pairs of ALL those cells -- lane L contributes end_L - i_L of them -- are laid out back to back
//      (a wave prefix sum) and tested 64 at a time by whichever lanes are free, each tester fetching "its" ray, maxt and cell window
//      from the owner's REGISTERS by ds_bpermute.  Hits go back to the owner through one LDS atomic: a 64-bit minimum over (t, primitive index).
// A lane's ray meets exactly the cells, and in each cell exactly the primitives with exactly the [cmin, cmax] windows, of the
// reference's nested loops (A10 code.cl:937-1070, 1195-1321); only WHO evaluates a test and in which order changes.  Order does not
// matter: inside a cell the reference keeps the hit with the smallest t, the first one among equals (strict <, code.cl:1017-1026) --
// the minimum of (t, index) -- and closes the walk at the first cell that produced one.  For shadow rays only "was anything hit with
// t < maxt" survives the fused kernel (sceneRender compares mint with maxt, code.cl:1339), so any hit will do there; the
// kernel-by-kernel path stores the shadow Ray and asks for the first hit in list order: the minimum of the index alone.
// t values are compared through the usual order-preserving map of IEEE bits to unsigned, after t + 0 (a -0 and a +0 are the same
// distance to the reference's <); the winner's own t / beta / gamma bits travel through three more LDS words.
//
// Requires EVERY lane of the wave to enter (lanes without a ray pass want = false): the prefix sums are DPP row operations.
#pragma once
#include "pt_trace.hpp"

#ifndef PT_COOP_REMAT
#define PT_COOP_REMAT 1
#endif

namespace pt {

// One wave's exchange area in the block's dynamic LDS: CW_ROWS rows of 64 words, [row][lane]; 1.5 KB per wave, 6 KB per block, at
// the start of the dynamic segment (the staged cell-offset tables follow: launch_fused).
// The owner's maxt, cell window and pair base are NOT rows: they reach a tester through ds_bpermute from the owner's registers (four
// rows = 4 KB of LDS per block less, and cornell_teapot3 853 -> 859 Msamples/s) -- and since round 3 neither is its ray (six rows more:
// what lets six blocks share a CU's LDS, pt_kernels_fused.hip PT_FUSED_WAVES_GRIDS).
#ifndef PT_COOP_RUNNING_CELL
#define PT_COOP_RUNNING_CELL 1   // phase A carries the cell index and the three slab indices (packed) along instead of re-deriving the index per step
#endif
#ifndef PT_COOP_RAY_BPERMUTE
#define PT_COOP_RAY_BPERMUTE 1   // the owner's ray reaches a tester by ds_bpermute from the owner's registers too: six rows (6 KB per block) less
#endif
#if PT_COOP_RAY_BPERMUTE
enum { CW_OWN = 0, CW_T, CW_BETA, CW_GAMMA, CW_KEY /* two rows: 64 x u64 */, CW_ROWS = 6 };
#else
enum { CW_OX = 0, CW_OY, CW_OZ, CW_DX, CW_DY, CW_DZ, CW_OWN, CW_T, CW_BETA, CW_GAMMA, CW_KEY /* two rows: 64 x u64 */, CW_ROWS = 12 };
#endif
constexpr uint32_t kCoopWordsPerBlock = 4u * CW_ROWS * 64u;

template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
PT_DEV uint32_t dpp_or0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, BOUND); }
// inclusive prefix sum / prefix maximum over the 64 lanes (all active): four shifts inside each row of 16, then the row totals
// carried to the next row and to the upper half (row_bcast:15, row_bcast:31)
PT_DEV uint32_t wave_scan_add(uint32_t v) {
    v += dpp_or0<0x111, 0xf, 0xf, true>(v);
    v += dpp_or0<0x112, 0xf, 0xf, true>(v);
    v += dpp_or0<0x114, 0xf, 0xf, true>(v);
    v += dpp_or0<0x118, 0xf, 0xf, true>(v);
    v += dpp_or0<0x142, 0xa, 0xf, false>(v);
    v += dpp_or0<0x143, 0xc, 0xf, false>(v);
    return v;
}
PT_DEV uint32_t umax2(uint32_t a, uint32_t b) { return a > b ? a : b; }
PT_DEV uint32_t wave_scan_max(uint32_t v) {
    v = umax2(v, dpp_or0<0x111, 0xf, 0xf, true>(v));
    v = umax2(v, dpp_or0<0x112, 0xf, 0xf, true>(v));
    v = umax2(v, dpp_or0<0x114, 0xf, 0xf, true>(v));
    v = umax2(v, dpp_or0<0x118, 0xf, 0xf, true>(v));
    v = umax2(v, dpp_or0<0x142, 0xa, 0xf, false>(v));
    v = umax2(v, dpp_or0<0x143, 0xc, 0xf, false>(v));
    return v;
}
// LDS traffic between lanes of one wave: the LDS executes a wave's instructions in order; this keeps the compiler from moving them
PT_DEV void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// t -> a key whose unsigned order is the order of the values; -0 counts as +0
PT_DEV uint32_t t_key(float t) {
    const uint32_t b = __float_as_uint(t + 0.0f);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}

// What a walk reports:
//   COOP_CLOSEST   the closest hit of the first cell that has one: index, t, beta, gamma                       (code.cl:937-1070)
//   COOP_ANY       whether anything is hit (idx != UINT32_MAX); t is not delivered -- the fused pass only asks "blocked?"
//   COOP_ANY_FIRST the reference's shadow loop to the letter: the FIRST primitive of the cell, in list order, that is hit, and its t
//                  (code.cl:1195-1321 leaves mint = maxt = that t in the shadow Ray, which the kernel-by-kernel path stores)
enum CoopMode { COOP_CLOSEST = 0, COOP_ANY = 1, COOP_ANY_FIRST = 2 };

template <int MODE, bool FAST, bool LDS_TABLES>
PT_DEV Hit trace_dda_coop(bool want, const Ray& ray, const BoxHit& bh, const GridArgs& S, bool& defer) {
    constexpr bool ANY = MODE == COOP_ANY;
    Hit ch;
    ch.idx = UINT32_MAX;
    ch.t = ray.maxt;
    ch.beta = 0.0f;
    ch.gamma = 0.0f;
    if (__builtin_amdgcn_ballot_w64(want) == 0ull) return ch;   // wave-uniform
    constexpr int PCK = MODE == COOP_CLOSEST ? 0 : 1;
    pt_count(PC_GRID_WALKS + PCK);
    if (PT_COUNT && want) pt_count(PC_GRID_WANT_LANES + PCK, true);
    const float4* __restrict__ prims = (const float4*)S.prims;
    const uint32_t* __restrict__ off = (const uint32_t*)S.off;
    uint32_t tid = threadIdx.x;
#if PT_COOP_REMAT
    asm volatile("" : "+v"(tid));   // the exchange-row addresses are re-derived per walk (three instructions) instead of being hoisted out of the
                                    // segment loop and kept alive -- in scratch, the one register the allocator had nowhere else to put
#endif
    const uint32_t lane = tid & 63u;
    const uint32_t wbase = (tid >> 6) * (CW_ROWS * 64u);
#define CW_MINE(row) pt_lds_dyn[wbase + (uint32_t)(row) * 64u + lane]
#define CW_OF(row, l) pt_lds_dyn[wbase + (uint32_t)(row) * 64u + (l)]
    unsigned long long* const keys = (unsigned long long*)&pt_lds_dyn[wbase + (uint32_t)CW_KEY * 64u];
    const unsigned long long kNone = ~0ull;

    float tnx = 0.0f, tny = 0.0f, tnz = 0.0f, dtx = 0.0f, dty = 0.0f, dtz = 0.0f;
    int sx = 0, sy = 0, sz = 0;
    const uint32_t zs = S.n * S.n, ys = S.n;   // n <= 1024 (check_grid): 24-bit multiplies are exact
#if PT_COOP_RUNNING_CELL
    const uint32_t last = S.n - 1u;
    uint32_t cell = 0u, pk = 0u;
#else
    const int nn = (int)S.n;
#endif
    const bool fwx = ray.d.x >= 0, fwy = ray.d.y >= 0, fwz = ray.d.z >= 0;   // code.cl:701-705: d >= 0 ? +1, n : -1, -1
    float cmin = 0.0f, cmax = 0.0f;
    uint32_t i = 0u, end = 0u;
    bool alive = want && !(bh.tmin >= ray.maxt);
    if (want) {
        bool dfr = FAST && S.walk_ok == 0u;
        const Axis ax = axis_setup_t<FAST>(ray.o.x, ray.d.x, bh.tmin, S.bound[0], S.bound[4], S.n, S.delta[0], S.rdelta[0], dfr);
        const Axis ay = axis_setup_t<FAST>(ray.o.y, ray.d.y, bh.tmin, S.bound[1], S.bound[5], S.n, S.delta[1], S.rdelta[1], dfr);
        const Axis az = axis_setup_t<FAST>(ray.o.z, ray.d.z, bh.tmin, S.bound[2], S.bound[6], S.n, S.delta[2], S.rdelta[2], dfr);
        defer = defer || dfr;
        tnx = ax.tnext; tny = ay.tnext; tnz = az.tnext;
        dtx = ax.dt; dty = ay.dt; dtz = az.dt;
        sx = ax.slab; sy = ay.slab; sz = az.slab;
        cmin = bh.tmin;
        cmax = cl_min(cl_min(tnx, tny), tnz);
#if PT_COOP_RUNNING_CELL
        cell = __umul24((uint32_t)sz, zs) + __umul24((uint32_t)sy, ys) + (uint32_t)sx;
        pk = (uint32_t)sx | (uint32_t)sy << 10 | (uint32_t)sz << 20;
        cell_range<LDS_TABLES>(S, off, cell, i, end);
#else
        cell_range<LDS_TABLES>(S, off, __umul24((uint32_t)sz, zs) + __umul24((uint32_t)sy, ys) + (uint32_t)sx, i, end);
#endif
#if !PT_COOP_RAY_BPERMUTE
        CW_MINE(CW_OX) = __float_as_uint(ray.o.x); CW_MINE(CW_OY) = __float_as_uint(ray.o.y); CW_MINE(CW_OZ) = __float_as_uint(ray.o.z);
        CW_MINE(CW_DX) = __float_as_uint(ray.d.x); CW_MINE(CW_DY) = __float_as_uint(ray.d.y); CW_MINE(CW_DZ) = __float_as_uint(ray.d.z);
#endif
    }
    keys[lane] = kNone;
    // one past the last slot of the set: a (ray, primitive) pair is only ever formed below it (a table that lies cannot send a load astray)
    uint32_t nslots = S.nslots;
    if (nslots == 0u) {
        uint32_t unused_;
        cell_range<LDS_TABLES>(S, off, zs * S.n - 1u, unused_, nslots);
        nslots = __builtin_amdgcn_readfirstlane(nslots);
    }

    for (;;) {
        // ---- phase A: through empty cells (code.cl:1028-1066's step, unchanged)
        pt_count(PC_GRID_PHASES + PCK);
        while (alive && i == end) {
            pt_count(PC_GRID_A_STEPS + PCK); pt_count(PC_GRID_A_LANE_STEPS + PCK, true);
            const float t = cmax;
            bool out;
#if PT_COOP_RUNNING_CELL
            // the slab indices packed ten bits each (0 <= slab < n <= 1024 while the ray is inside) and the cell index carried along:
            // "the step leaves the grid" (code.cl:701-705's limit, n or -1) is "the slab stepped FROM is n - 1 or 0"
            if (t == tnx) {
                tnx += dtx;
                out = t >= bh.tmax || (pk & 1023u) == (fwx ? last : 0u);
                pk += fwx ? 1u : ~0u;
                cell += fwx ? 1u : ~0u;
            } else if (t == tny) {
                tny += dty;
                out = t >= bh.tmax || ((pk >> 10) & 1023u) == (fwy ? last : 0u);
                pk += fwy ? 1u << 10 : 0u - (1u << 10);
                cell += fwy ? ys : 0u - ys;
            } else {
                tnz += dtz;
                out = t >= bh.tmax || (pk >> 20) == (fwz ? last : 0u);
                pk += fwz ? 1u << 20 : 0u - (1u << 20);
                cell += fwz ? zs : 0u - zs;
            }
#else
            if (t == tnx) {
                tnx += dtx;
                sx += fwx ? 1 : -1;
                out = t >= bh.tmax || sx == (fwx ? nn : -1);
            } else if (t == tny) {
                tny += dty;
                sy += fwy ? 1 : -1;
                out = t >= bh.tmax || sy == (fwy ? nn : -1);
            } else {
                tnz += dtz;
                sz += fwz ? 1 : -1;
                out = t >= bh.tmax || sz == (fwz ? nn : -1);
            }
#endif
            // ... or the cell starts at or beyond the ray's end: a hit needs cmin <= t < maxt, and cmin only grows from here
            // (the reference walks on to the grid's far side rejecting every hit; nothing it computes there survives)
            if (out || t >= ray.maxt) { alive = false; break; }
            cmin = t;
            cmax = cl_min(cl_min(tnx, tny), tnz);
#if PT_COOP_RUNNING_CELL
            cell_range<LDS_TABLES>(S, off, cell, i, end);
#else
            cell_range<LDS_TABLES>(S, off, __umul24((uint32_t)sz, zs) + __umul24((uint32_t)sy, ys) + (uint32_t)sx, i, end);
#endif
        }
        if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;
        // ---- phase B: every pair (owner lane, primitive of its cell), 64 at a time
        const uint32_t cnt = alive ? end - i : 0u;
        const uint32_t incl = wave_scan_add(cnt), excl = incl - cnt;
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        pt_count(PC_GRID_PAIRS + PCK, false, total);
        for (uint32_t base = 0u; base < total; base += 64u) {
            pt_count(PC_GRID_ROUNDS + PCK);
            // who owns pair base + lane: owners mark the first pair of theirs inside this window, a prefix maximum spreads the mark
            CW_MINE(CW_OWN) = 0u;
            if (cnt != 0u && excl < base + 64u && incl > base) CW_OF(CW_OWN, (excl > base ? excl : base) - base) = lane + 1u;
            wave_fence();
            const uint32_t mark = wave_scan_max(CW_MINE(CW_OWN));
            const uint32_t p = base + lane;
            // pair p of the wave is primitive (i - excl) + p of its owner's set.  Every lane asks (a lane without a pair asks lane 0 and drops the answer): ds_bpermute reads the owners' registers
            const int oaddr = (int)((mark != 0u ? mark - 1u : 0u) << 2);
            const float omax = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(oaddr, (int)__float_as_uint(ray.maxt)));
            const float ocmin = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(oaddr, (int)__float_as_uint(cmin)));
            const float ocmax = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(oaddr, (int)__float_as_uint(cmax)));
            const uint32_t oibx = (uint32_t)__builtin_amdgcn_ds_bpermute(oaddr, (int)(i - excl));
#if PT_COOP_RAY_BPERMUTE
            auto pull = [&](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(oaddr, (int)__float_as_uint(v))); };
            const f3 ro = mk3(pull(ray.o.x), pull(ray.o.y), pull(ray.o.z));
            const f3 rd = mk3(pull(ray.d.x), pull(ray.d.y), pull(ray.d.z));
#endif
            if (p < total && mark != 0u) {
                const uint32_t o = mark - 1u;
                const uint32_t prim = oibx + p;
                if (prim < nslots) {
#if !PT_COOP_RAY_BPERMUTE
                    const f3 ro = mk3(__uint_as_float(CW_OF(CW_OX, o)), __uint_as_float(CW_OF(CW_OY, o)), __uint_as_float(CW_OF(CW_OZ, o)));
                    const f3 rd = mk3(__uint_as_float(CW_OF(CW_DX, o)), __uint_as_float(CW_OF(CW_DY, o)), __uint_as_float(CW_OF(CW_DZ, o)));
#endif
                    const float4* __restrict__ q = prims + 3u * (size_t)prim;
                    float tt, bb, gg;
                    if (tri_test<TRI_A10, FAST>(ro, rd, ocmin, ocmax, q[0], q[1], q[2], tt, bb, gg) && tt < omax) {
                        if (ANY) keys[o] = 0ull;
                        else {
                            // COOP_ANY_FIRST: the lowest primitive index wins whatever its t
                            const unsigned long long mk = MODE == COOP_ANY_FIRST ? (unsigned long long)prim : (((unsigned long long)t_key(tt) << 32) | prim);
                            __hip_atomic_fetch_min(&keys[o], mk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            // the pair that holds its owner's minimum hands over its own t / beta / gamma bits.  Every lane's minimum of
                            // this round is in before any lane's read below (one wave: the LDS runs its instructions in order)
                            wave_fence();
                            if (keys[o] == mk) {
                                CW_OF(CW_T, o) = __float_as_uint(tt);
                                CW_OF(CW_BETA, o) = __float_as_uint(bb);
                                CW_OF(CW_GAMMA, o) = __float_as_uint(gg);
                            }
                        }
                    }
                }
            }
        }
        wave_fence();
        // the cell is done: a hit inside it ends the walk (code.cl:768-771), else the DDA steps on
        if (alive) {
            if (keys[lane] != kNone) alive = false;
            else i = end;
        }
    }
    if (want) {
        const unsigned long long k = keys[lane];
        if (k != kNone) {
            ch.idx = (uint32_t)k;
            if (!ANY) {
                ch.t = __uint_as_float(CW_MINE(CW_T));
                if (MODE == COOP_CLOSEST) {
                    ch.beta = __uint_as_float(CW_MINE(CW_BETA));
                    ch.gamma = __uint_as_float(CW_MINE(CW_GAMMA));
                }
            }
        }
    }
#undef CW_MINE
#undef CW_OF
    return ch;
}

}  // namespace pt
