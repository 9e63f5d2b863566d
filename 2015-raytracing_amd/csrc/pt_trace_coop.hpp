// pt_trace_coop.hpp -- the grid walk of the fused pass with the triangle tests SHARED by the wave.
//
// trace_dda (pt_trace.hpp) lets every lane test one triangle of its own cell per trip.  On an incoherent wave (bounce and shadow rays
// against a mesh in an n^3 grid) a quarter of the lanes hold a triangle at any time, and a 60-instruction test issues for the whole
// wave however few do: the walk is VALU-issue-bound at 24 % lane utilisation (cornell_teapot3).
//
// Here a walk alternates two wave-wide phases:
//   A  every live lane steps its own DDA along its WHOLE way through the grid -- to where the ray leaves it, or to the first cell that starts
//      at or beyond the ray's end -- and leaves one record {first slot, count, cmin, cmax} per cell that holds primitives in the wave's
//      record pool (64 records in LDS; a lane that finds the pool full keeps its cell for the next sweep);
//   B  the (ray, primitive) pairs of ALL those records are laid out back to back (a wave prefix sum over the records' counts) and tested 64
//      at a time by whichever lanes are free, each tester fetching "its" ray and its end from the owner's REGISTERS by ds_bpermute and the
//      cell's window from the record.  Hits go back to the owner through one LDS atomic: a 64-bit minimum over (t, record, slot in the cell).
// Round 3's walk pooled one cell per lane and phase: the lanes of a wave need different numbers of cells (one and a half on average, five
// for the slowest of a wave), so a walk took five phases whose later rounds tested a handful of pairs -- 23 of 64 on average (profiles/
// r4_work/trips_teapot3_r3walk.txt).  With the whole way pooled a walk is one sweep, seldom two.
//
// A lane's ray meets exactly the cells, and in each cell exactly the primitives with exactly the [cmin, cmax] windows, of the reference's
// nested loops (A10 code.cl:937-1070, 1195-1321) -- plus, for a ray that hits, the cells BEHIND the hit the reference never reaches.
// Nothing they yield survives: the reference closes the walk at the first cell that produced a hit and keeps, inside it, the hit with the
// smallest t, the first one among equals (strict <, code.cl:1017-1026); cells are met in increasing t (a later cell's window starts where
// the earlier one ends, and a hit lies inside its cell's window), so that hit IS the minimum of (t, record, slot) over everything the lane
// pooled: a later cell can tie on t at best, and then loses on its record number (a lane's records are numbered in the order it met the
// cells).  For shadow rays only "was anything hit with t < maxt" survives the fused kernel (sceneRender compares mint with maxt,
// code.cl:1339), so any hit will do there; the kernel-by-kernel path stores the shadow Ray and asks for the first hit in list order of the
// first cell that has one: the minimum of (record, slot) alone.  t values are compared through the usual order-preserving map of IEEE bits
// to unsigned, after t + 0 (a -0 and a +0 are the same distance to the reference's <).  The winner's t / beta / gamma are not passed back:
// its owner runs the reference's test on it once more -- same operands, same operations, same bits -- which costs one test per walk and
// saves three exchange rows per wave (the 768 bytes that keep six blocks on a CU).
//
// How far ahead of the verdict a lane pools (measured with a cap on the cells per sweep, cornell_teapot3 / cornell_teapot 1080p x 16, ms per pass):
// 1 / 2 / 3 / 4 / its whole way = 33.6 / 28.9 / 27.1 / 26.5 / 26.4 and 23.2 / 20.0 / 18.8 / 18.4 / 18.2 (round 3's one-cell walk without a pool:
// 29.1 / 19.9).  A ray that hits in its first cell has no use for the cells behind it -- the whole way tests 54 % more pairs than the reference
// makes -- and still it is the sweeps that cost, not the pairs.
//
// Requires EVERY lane of the wave to enter (lanes without a ray pass want = false): the prefix sums are DPP row operations.
#pragma once
#include "pt_trace.hpp"

#ifndef PT_COOP_STEP_ASM
#define PT_COOP_STEP_ASM 1   // phase A's stepping loop with its control flow written out (0: the same loop left to the compiler; PT_COUNT builds and tables in memory use that one)
#endif
#ifndef PT_COOP_REMAT
#define PT_COOP_REMAT 1
#endif

namespace pt {

// One wave's exchange area in the block's dynamic LDS, [row][lane] rows of 64 words at the start of the dynamic segment (the staged
// cell-offset tables follow: launch_fused): the owner marks of a round, the 64 keys (two rows), the record pool (64 records of four words:
// four rows) and the pool's counter.  The owner's ray and end reach a tester through ds_bpermute from the owner's registers.
enum { CW_OWN = 0, CW_KEY /* two rows: 64 x u64 */, CW_POOL = 3 /* four rows: 64 x {first, count | owner << 24, cmin, cmax} */, CW_COUNT = 7 /* one word (a 16-byte row) */ };
constexpr uint32_t kCoopWordsPerWave = 7u * 64u + 4u;
constexpr uint32_t kCoopWordsPerBlock = 4u * kCoopWordsPerWave;
constexpr uint32_t kCoopPool = 64;               // records per sweep
constexpr uint32_t kCoopMaxCell = kMaxCellSlots; // slots one cell may hold (the key packs a slot's place in its cell into 24 bits): check_grid refuses more

template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
PT_DEV uint32_t dpp_or0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, BOUND); }
// inclusive prefix sum over the 64 lanes (all active): four shifts inside each row of 16, then the row totals
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
PT_DEV Hit trace_dda_coop(bool want, const Ray& ray, const RayRcp& rr, const BoxHit& bh, const GridArgs& S, bool& defer) {
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
    const uint32_t wbase = __umul24(tid >> 6, kCoopWordsPerWave);
#define CW_MINE(row) pt_lds_dyn[wbase + (uint32_t)(row) * 64u + lane]
#define CW_OF(row, l) pt_lds_dyn[wbase + (uint32_t)(row) * 64u + (l)]
    unsigned long long* const keys = (unsigned long long*)&pt_lds_dyn[wbase + (uint32_t)CW_KEY * 64u];
    uint4* const pool = (uint4*)&pt_lds_dyn[wbase + (uint32_t)CW_POOL * 64u];
    uint32_t* const pool_count = &pt_lds_dyn[wbase + (uint32_t)CW_COUNT * 64u];
    const unsigned long long kNone = ~0ull;

    // the staged table's place in the block's LDS, pinned in a scalar register: left to itself the compiler, short of SGPRs, re-loaded it from the
    // kernel arguments inside the stepping loop -- a scalar-memory round trip ahead of every step's LDS read
    uint32_t toff = S.lds_off;
    asm volatile("" : "+s"(toff));
    auto range = [&](uint32_t c, uint32_t& first, uint32_t& past) {
        if (LDS_TABLES) { first = pt_lds_dyn[toff + c]; past = pt_lds_dyn[toff + c + 1u]; }
        else { first = off[c]; past = off[c + 1u]; }
    };
    float tnx = 0.0f, tny = 0.0f, tnz = 0.0f, dtx = 0.0f, dty = 0.0f, dtz = 0.0f;
    const uint32_t zs = S.n * S.n, ys = S.n;   // n <= 1024 (check_grid): 24-bit multiplies are exact
    const uint32_t last = S.n - 1u;
    uint32_t cell = 0u, pk = 0u;
    uint32_t scx = 0u, scy = 0u, scz = 0u;   // what a step along x / y / z adds to the cell index: +-1, +-n, +-n^2
    const bool fwx = ray.d.x >= 0, fwy = ray.d.y >= 0, fwz = ray.d.z >= 0;   // code.cl:701-705: d >= 0 ? +1, n : -1, -1
    float cmin = 0.0f, cmax = 0.0f;
    uint32_t i = 0u, end = 0u;
    bool alive = want && !(bh.tmin >= ray.maxt);
    // where the walk ends at the latest: the ray's way out of the set's box, or its own end (t >= tmax || t >= maxt is t >= the smaller of the two:
    // v_min_f32 passes the other operand when one is a NaN, and a comparison with a NaN is false either way)
    const float tend = cl_min(bh.tmax, ray.maxt);
    if (want) {
        bool dfr = FAST && S.walk_ok == 0u;
        const Axis ax = axis_setup_t<FAST>(ray.o.x, ray.d.x, bh.tmin, S.bound[0], S.bound[4], S.n, S.delta[0], S.rdelta[0], rr.x, dfr);
        const Axis ay = axis_setup_t<FAST>(ray.o.y, ray.d.y, bh.tmin, S.bound[1], S.bound[5], S.n, S.delta[1], S.rdelta[1], rr.y, dfr);
        const Axis az = axis_setup_t<FAST>(ray.o.z, ray.d.z, bh.tmin, S.bound[2], S.bound[6], S.n, S.delta[2], S.rdelta[2], rr.z, dfr);
        defer = defer || dfr;
        tnx = ax.tnext; tny = ay.tnext; tnz = az.tnext;
        dtx = ax.dt; dty = ay.dt; dtz = az.dt;
        cmin = bh.tmin;
        cmax = cl_min(cl_min(tnx, tny), tnz);
        // the cell index is carried along
        cell = __umul24((uint32_t)az.slab, zs) + __umul24((uint32_t)ay.slab, ys) + (uint32_t)ax.slab;
        // pk: the steps LEFT on each axis before the ray leaves the grid, ten bits each (0 <= slab < n <= 1024 while the ray is inside): "the step leaves
        // the grid" (code.cl:701-705's limit, n or -1) is "the slab stepped FROM is n - 1 or 0" is "the stepped axis' field is 0"
        pk = (fwx ? last - (uint32_t)ax.slab : (uint32_t)ax.slab) | (fwy ? last - (uint32_t)ay.slab : (uint32_t)ay.slab) << 10 | (fwz ? last - (uint32_t)az.slab : (uint32_t)az.slab) << 20;
        scx = fwx ? 1u : ~0u; scy = fwy ? ys : 0u - ys; scz = fwz ? zs : 0u - zs;
        range(cell, i, end);
    }
    {   // keys[lane] = kNone, from a constant made here: hoisted out of the segment loop the pair of all-ones registers was kept in scratch and re-loaded per walk
        uint32_t ones = ~0u;
        asm volatile("" : "+v"(ones));
        keys[lane] = (unsigned long long)ones << 32 | ones;
    }
    // one past the last slot of the set: a (ray, primitive) pair is only ever formed below it (a table that lies cannot send a load astray)
    uint32_t nslots = S.nslots;
    if (nslots == 0u) {
        uint32_t unused_;
        range(zs * S.n - 1u, unused_, nslots);
        nslots = __builtin_amdgcn_readfirstlane(nslots);
    }

    for (;;) {
        if (lane == 0u) *pool_count = 0u;
        wave_fence();
        pt_count(PC_GRID_PHASES + PCK);
        // ---- phase A: the lane's whole way through the grid (code.cl:1028-1066's step), one record per cell with a list.
        // The step as selects, on a packing made for it: the stepped axis' steps-left field is one variable-offset v_bfe_u32 and its update always a
        // subtraction, the cell index moves by a per-lane constant.  Measured (cornell_teapot3 / cornell_teapot / own_gems 1080p x 16, ms per pass,
        // profiles/r4_ab.tsv): the reference's if / else-if / else with slab indices 24.2 / 16.8 / 11.5; this, compiled, 23.6 / 16.1 / 10.9; this with
        // the loop's control flow written out (below) 21.0 / 14.5 / 10.4.
        // Measured and not kept: the walk's state kept out of phase B by walking the way again after a full pool (no spill
        // in the 96-register build, 5 % slower: pools do fill); record slots handed out by a ballot instead of the LDS atomic (bit-identical,
        // 10 % slower: the compiler's exec-mask bookkeeping for the loop grew by a quarter); "anything ahead?" bits per cell and octant that
        // end a walk with only empty cells left (-18 % lane steps, 5 % slower: one more test and register in this loop).
        if (PT_COOP_STEP_ASM && !PT_COUNT && LDS_TABLES) {
#if PT_COOP_STEP_ASM
            if (alive) {
            // The loop of the other branch, operation for operation, with the control flow the compiler cannot express: its structurizer spends 27
            // scalar instructions a step on the exec masks of a loop with two exits and a rare side path (beside 23 vector instructions of
            // work); here a step is 22 vector + 6 scalar instructions + 2 LDS reads, the reads issued as soon as the next cell is known.  exec on
            // entry = the lanes that walk; a lane leaves by clearing its exec bit (its walk is over) or through `stall` (it found the pool full and
            // walks on in the next sweep).  Hazards: none of CDNA3's manually handled ones occur (no DPP, no v_readlane, no VMEM after a VALU-written
            // SGPR); the compiler itself overwrites an LDS address register right behind the ds_read that used it.
            uint32_t flag, tb, tc, tsh;
            unsigned long long s_save, s_stall, s_ex, s_ey, s_t, s_in;
            const uint32_t pool_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&pt_lds_dyn[wbase + (uint32_t)CW_POOL * 64u];
            const uint32_t tbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&pt_lds_dyn[toff];
            const uint32_t lane24 = lane << 24;
            static_assert(((uint32_t)CW_COUNT - (uint32_t)CW_POOL) * 64u * 4u == 1024u, "the pool's counter sits 1024 bytes behind its first record");
            asm volatile(
                "s_mov_b64 %[save], exec\n\t"
                "s_mov_b64 %[stall], 0\n"
                ".Lcoop_top_%=:\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_cmp_ne_u32_e32 vcc, %[i], %[end]\n\t"
                "s_cbranch_vccz .Lcoop_step_%=\n\t"
                // lanes whose cell holds primitives: a slot of the pool each, or -- pool full -- out of this sweep with the cell still theirs
                "s_mov_b64 %[t], exec\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "v_mov_b32_e32 %[a], 1\n\t"
                "ds_add_rtn_u32 %[a], %[pool], %[a] offset:1024\n\t"
                "v_sub_u32_e32 %[b], %[end], %[i]\n\t"
                "v_or_b32_e32 %[b], %[b], %[lane24]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_cmp_gt_u32_e32 vcc, 64, %[a]\n\t"
                "s_andn2_b64 %[ex], exec, vcc\n\t"
                "s_or_b64 %[stall], %[stall], %[ex]\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "v_lshl_add_u32 %[a], %[a], 4, %[pool]\n\t"
                "ds_write2_b32 %[a], %[i], %[b] offset1:1\n\t"
                "ds_write2_b32 %[a], %[cmin], %[cmax] offset0:2 offset1:3\n\t"
                "s_andn2_b64 exec, %[t], %[ex]\n\t"
                "s_cbranch_execz .Lcoop_done_%=\n"
                ".Lcoop_step_%=:\n\t"
                // the step (code.cl:1028-1066): the axis whose plane the ray reached -- x before y before z -- moves on
                "v_cmp_eq_f32_e64 %[ex], %[cmax], %[tnx]\n\t"
                "v_cmp_eq_f32_e64 %[ey], %[cmax], %[tny]\n\t"
                "v_cmp_nge_f32_e32 vcc, %[cmax], %[tend]\n\t"
                "s_or_b64 %[t], %[ex], %[ey]\n\t"
                "s_andn2_b64 %[ey], %[ey], %[ex]\n\t"
                "v_cndmask_b32_e64 %[sh], 20, 10, %[ey]\n\t"
                "v_cndmask_b32_e64 %[sh], %[sh], 0, %[ex]\n\t"
                "v_cndmask_b32_e64 %[a], %[scz], %[scy], %[ey]\n\t"
                "v_bfe_u32 %[b], %[pk], %[sh], 10\n\t"
                "v_cndmask_b32_e64 %[a], %[a], %[scx], %[ex]\n\t"
                "v_cmp_ne_u32_e64 %[save2], 0, %[b]\n\t"
                "v_add_u32_e32 %[cell], %[cell], %[a]\n\t"
                "s_and_b64 vcc, vcc, %[save2]\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz .Lcoop_done_%=\n\t"
                // the next cell's list: asked for before the rest of the step, which does not need it
                "v_lshl_add_u32 %[a], %[cell], 2, %[tbase]\n\t"
                "ds_read_b32 %[i], %[a]\n\t"
                "ds_read_b32 %[end], %[a] offset:4\n\t"
                "v_add_f32_e32 %[a], %[tnx], %[dtx]\n\t"
                "v_add_f32_e32 %[b], %[tny], %[dty]\n\t"
                "v_add_f32_e32 %[c], %[tnz], %[dtz]\n\t"
                "v_lshl_add_u32 %[pk], -1, %[sh], %[pk]\n\t"
                "v_mov_b32_e32 %[cmin], %[cmax]\n\t"
                "v_cndmask_b32_e64 %[tnx], %[tnx], %[a], %[ex]\n\t"
                "v_cndmask_b32_e64 %[tny], %[tny], %[b], %[ey]\n\t"
                "v_cndmask_b32_e64 %[tnz], %[c], %[tnz], %[t]\n\t"
                "v_min3_f32 %[cmax], %[tnx], %[tny], %[tnz]\n\t"
                "s_branch .Lcoop_top_%=\n"
                ".Lcoop_done_%=:\n\t"
                "s_mov_b64 exec, %[save]\n\t"
                "v_cndmask_b32_e64 %[a], 0, 1, %[stall]"
                : [tnx] "+v"(tnx), [tny] "+v"(tny), [tnz] "+v"(tnz), [pk] "+v"(pk), [cell] "+v"(cell), [cmin] "+v"(cmin), [cmax] "+v"(cmax), [i] "+v"(i), [end] "+v"(end),
                  [a] "=&v"(flag), [b] "=&v"(tb), [c] "=&v"(tc), [sh] "=&v"(tsh),
                  [save] "=&s"(s_save), [stall] "=&s"(s_stall), [ex] "=&s"(s_ex), [ey] "=&s"(s_ey), [t] "=&s"(s_t), [save2] "=&s"(s_in)
                : [dtx] "v"(dtx), [dty] "v"(dty), [dtz] "v"(dtz), [scx] "v"(scx), [scy] "v"(scy), [scz] "v"(scz), [tend] "v"(tend), [lane24] "v"(lane24),
                  [pool] "v"(pool_addr), [tbase] "s"(tbase)
                : "vcc", "scc", "memory");
            alive = flag != 0u;
            }
#endif
        } else {
            bool stalled = false;   // "found the pool full": the only way a lane leaves the loop below and walks on in the next sweep
            if (alive) for (;;) {
                pt_count(PC_GRID_A_STEPS + PCK); pt_count(PC_GRID_A_LANE_STEPS + PCK, true);
                if (__builtin_expect(i != end, 0)) {
                    const uint32_t slot = __hip_atomic_fetch_add(pool_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    if (slot >= kCoopPool) { stalled = true; break; }   // the pool is full: this cell is the first record of the lane's next sweep
                    pool[slot] = make_uint4(i, (end - i) | lane << 24, __float_as_uint(cmin), __float_as_uint(cmax));
                }
                const float t = cmax;
                // the reference's if / else-if / else (x, then y, then z: code.cl:1028-1066)
                const bool ex = t == tnx, ey = !ex && t == tny, ez = !ex && !ey;
                const float vx = tnx + dtx, vy = tny + dty, vz = tnz + dtz;
                tnx = ex ? vx : tnx; tny = ey ? vy : tny; tnz = ez ? vz : tnz;
                const uint32_t sh = ex ? 0u : (ey ? 10u : 20u);
                // the walk ends where the step leaves the grid, or where the next cell would start at or beyond the ray's end: a hit needs cmin <= t < maxt,
                // and cmin only grows from here (the reference walks on to the grid's far side rejecting every hit; nothing it computes there survives)
                const bool out = t >= tend || __builtin_amdgcn_ubfe(pk, sh, 10u) == 0u;
                pk -= 1u << sh;
                cell += ex ? scx : (ey ? scy : scz);
                if (out) break;
                cmin = t;
                cmax = cl_min(cl_min(tnx, tny), tnz);
                range(cell, i, end);
            }
            alive = stalled;
        }
        wave_fence();
        uint32_t n_rec = __builtin_amdgcn_readfirstlane(*pool_count);
        if (n_rec == 0u) break;   // (no record: every lane's walk is over)
        n_rec = n_rec < kCoopPool ? n_rec : kCoopPool;
        // ---- phase B: every pair (record, slot of its cell), 64 at a time.  Lane r speaks for record r.
        const uint32_t cnt = lane < n_rec ? pool[lane].y & (kCoopMaxCell - 1u) : 0u;
        const uint32_t incl = wave_scan_add(cnt), excl = incl - cnt;
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        pt_count(PC_GRID_PAIRS + PCK, false, total);
        for (uint32_t base = 0u; base < total; base += 64u) {
            pt_count(PC_GRID_ROUNDS + PCK);
            // which record pair base + lane belongs to: records mark the first pair of theirs inside this window; the records of a window are
            // consecutive, so a pair's record is the one that marked pair `base` plus the marks between the two -- one ballot and a v_mbcnt pair
            // (a DPP prefix maximum over the marks before: six dependent DPP operations and their wait states per round)
            CW_MINE(CW_OWN) = 0u;
            if (cnt != 0u && excl < base + 64u && incl > base) CW_OF(CW_OWN, (excl > base ? excl : base) - base) = lane + 1u;
            wave_fence();
            const uint32_t own_mark = CW_MINE(CW_OWN);
            const bool marked = own_mark != 0u;
            const unsigned long long marks = __builtin_amdgcn_ballot_w64(marked);
            const uint32_t first_rec = (uint32_t)__builtin_amdgcn_readfirstlane((int)own_mark) - 1u;   // (pair `base` exists: its record marked lane 0)
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(marks >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)marks, 0u));
            const uint32_t p = base + lane;
            const uint32_t r = (first_rec - 1u + below + (marked ? 1u : 0u)) & 63u;   // (a lane past the last pair asks some record and drops the answers)
            const uint32_t rexcl = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(r << 2), (int)excl);
            const uint4 rec = pool[r];
            const uint32_t o = rec.y >> 24;
            const int oaddr = (int)(o << 2);
            auto pull = [&](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(oaddr, (int)__float_as_uint(v))); };
            const float omax = pull(ray.maxt);
            const f3 ro = mk3(pull(ray.o.x), pull(ray.o.y), pull(ray.o.z));
            const f3 rd = mk3(pull(ray.d.x), pull(ray.d.y), pull(ray.d.z));
            if (p < total) {
                const uint32_t j = p - rexcl;
                const uint32_t prim = rec.x + j;
                if (prim < nslots) {
                    const float4* __restrict__ q = prims + 3u * (size_t)prim;
                    float tt, bb, gg;
                    if (tri_test<TRI_A10, FAST>(ro, rd, __uint_as_float(rec.z), __uint_as_float(rec.w), q[0], q[1], q[2], tt, bb, gg) && tt < omax) {
                        if (ANY) keys[o] = 0ull;
                        else {
                            // (record, slot): a lane's records are numbered in the order it met the cells, so the first cell wins a tie on t and, inside
                            // a cell, the lower slot; COOP_ANY_FIRST: the first cell with a hit, its lowest slot, whatever the t
                            const uint32_t lo = r << 24 | j;
                            const unsigned long long mk = MODE == COOP_ANY_FIRST ? (unsigned long long)lo : (((unsigned long long)t_key(tt) << 32) | lo);
                            __hip_atomic_fetch_min(&keys[o], mk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    }
                }
            }
        }
        wave_fence();
        // a hit anywhere on the lane's way ends its walk (code.cl:768-771: the first cell that produced one closes it)
        const unsigned long long k = keys[lane];
        const bool got = want && k != kNone;
        if (got) alive = false;
        if (!ANY) {
            // the winner: its slot and its cell's window, out of the record while the pool still holds it (the window parks in beta / gamma)
            if (got && ch.idx == UINT32_MAX) {
                const uint32_t lo = (uint32_t)k;
                const uint4 rec = pool[lo >> 24];
                ch.idx = rec.x + (lo & (kCoopMaxCell - 1u));
                ch.beta = __uint_as_float(rec.z);
                ch.gamma = __uint_as_float(rec.w);
            }
        } else if (got) ch.idx = 0u;
        if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;
    }
    if (!ANY && __builtin_amdgcn_ballot_w64(ch.idx != UINT32_MAX) != 0ull) {
        // the winner once more, by its owner: the reference's test on the same operands gives the same t / beta / gamma
        const bool got = ch.idx != UINT32_MAX;
        if (got && ch.idx < nslots) {
            const float4* __restrict__ q = prims + 3u * (size_t)ch.idx;
            float tt, bb, gg;
            (void)tri_test<TRI_A10, FAST>(ray.o, ray.d, ch.beta, ch.gamma, q[0], q[1], q[2], tt, bb, gg);
            ch.t = tt;
            ch.beta = bb;
            ch.gamma = gg;
        } else if (got) ch.idx = UINT32_MAX;   // (a table that lies: no pair was formed at or beyond nslots, so no key can name one)
    }
#undef CW_MINE
#undef CW_OF
    return ch;
}

}  // namespace pt
