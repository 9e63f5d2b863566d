// mirt_abi.cpp -- implementation of include/mirt.h: contexts, buffers, the WebCL-shaped
// kernel objects (argument marshalling + launch validation) and the fused pass.
//
// Everything a launch will touch is checked on the host BEFORE the kernel is enqueued:
// buffer extents against the work-item count, cell-offset tables for monotonicity and
// extent, argument sizes.  A bad argument therefore comes back as an error code; it does
// not become an out-of-bounds access on the GPU.
#include "../../include/mirt.h"
#include "pt_launch.hpp"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <atomic>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

constexpr size_t kRayBytes = 48, kPoiBytes = 64, kAcuBytes = 16;

thread_local std::string t_last_error = "";

// Live handles, by address AND kind: a handle of the wrong type (a kernel passed where a buffer is expected, a stale address that
// `new` has since handed to another kind of object) fails validation instead of being dereferenced as something it is not.
enum HandleKind : uint8_t { H_CTX = 1, H_BUF, H_KERNEL, H_GRAPH, H_GROUP };
std::mutex g_live_mu;
std::unordered_map<const void*, HandleKind> g_live;
std::atomic<uint64_t> g_next_uid{1};

void live_add(const void* p, HandleKind k) { std::lock_guard<std::mutex> l(g_live_mu); g_live[p] = k; }
void live_del(const void* p) { std::lock_guard<std::mutex> l(g_live_mu); g_live.erase(p); }
bool live_is(const void* p, HandleKind k) {
    if (!p) return false;
    std::lock_guard<std::mutex> l(g_live_mu);
    auto it = g_live.find(p);
    return it != g_live.end() && it->second == k;
}

}  // namespace

struct mirt_graph;
struct mirt_group;

struct mirt_buf;
enum ArgType { A_BUF, A_U32, A_F32, A_F16, A_AABB };
enum KernelId {
    K_sizeofRay, K_sizeofPoi, K_initAcu, K_initTrace, K_sphereTrace, K_triangleTrace, K_meshTrace, K_lightRender,
    K_initShadowTrace, K_sphereShadowTrace, K_triangleShadowTrace, K_sceneRender, K_bouncePaths, K_copyToPixel,
    K_a01_raytrace, K_a04_sizeofRay, K_a04_initTrace, K_a04_meshTrace, K_a07_sizeofRay, K_a07_initTrace, K_a07_meshTrace, K_a07_molTrace, K_COUNT
};

struct KernelSpec { const char* name; KernelId id; std::vector<ArgType> args; };

struct KArg {
    bool set = false;
    mirt_buf* buf = nullptr;
    union { uint32_t u; float f; float v[16]; } val;
};

struct mirt_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string last_error;
    void* scratch = nullptr;  // lens draws of the rpp==1 mode
    size_t scratch_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void* defer = nullptr;        // optimistic pass: [count, cursor, pad, pad | mask words...]
    size_t defer_bytes = 0;
    uint32_t defer_words = 0;     // mask words the last mirt_render_pass used (0: it ran the exact kernel only)
    uint32_t defer_unit = 1;      // samples per mask bit: 1, or 256 when that pass resolved its pixels itself (a bit per block)
    bool inpass_resolve = true;   // a pass writes pixel / radiance itself where it can (MIRT_INPASS_RESOLVE=0: always the separate copyToPixel)
    bool last_pass_resolved = false;   // render_pass_impl: the pass just queued resolved its own pixels
    float res_m_override = NAN;        // try_fuse_pass -> render_pass_impl: the tone factor of the recorded copyToPixel, as the host passed it
    int force_exact = 0;          // mirt_ctx_set_exact_only: 1 = skip the optimistic kernel, run every sample through the exact one
    bool profiling = false;       // per-kernel events inside mirt_render_pass
    hipEvent_t pe[3] = {nullptr, nullptr, nullptr};
    bool pe_valid = false;
    bool capturing = false;       // between mirt_capture_begin and mirt_capture_end: the stream records, nothing may wait on it
    // objects created on this context; mirt_ctx_destroy reaps what the host did not release (the reference host leaks its
    // bouncePaths kernel: A10 code.js:1444-1455 never pushes it on cl_resources)
    std::unordered_set<mirt_buf*> bufs;
    std::unordered_set<mirt_kernel*> kernels;
    std::unordered_set<mirt_graph*> graphs;
    mirt_group* group = nullptr;  // set when the context belongs to a device group (mirt_group_create)
    // command-stream fusion (mirt_ctx_set_fusion): enqueues of the Assign10 pass kernels held back until the pass is complete
    int fusion = 0;
    struct Pending { const KernelSpec* spec; std::vector<KArg> args; unsigned dim; size_t g[3]; };
    std::vector<Pending> pending;
    uint64_t fused_passes = 0;    // passes executed as ONE fused launch because their enqueue stream matched executeRender's
    // what a recording has touched so far (mirt_graph pins): device allocations a replay will read or write
    uint64_t scratch_gen = 1, defer_gen = 1;   // bumped whenever the allocation is replaced
    struct Pin { mirt_buf* buf; uint64_t uid; uint64_t prep_gen; uint64_t version; bool content; };
    std::vector<Pin> cap_pins;
    bool cap_scratch = false, cap_defer = false;
};

// held-back enqueues (command-stream fusion, below): every entry point that observes or changes device state runs them first
static int flush_pending(mirt_ctx* ctx);
#define FLUSH_PENDING(ctx) do { if (!(ctx)->pending.empty()) { int _rc = flush_pending(ctx); if (_rc) return _rc; } } while (0)

struct mirt_graph {
    mirt_ctx* ctx = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // every allocation the recording captured a raw pointer of; mirt_graph_launch refuses to replay once one of them is gone
    std::vector<mirt_ctx::Pin> pins;
    uint64_t scratch_gen = 0, defer_gen = 0;   // 0: not used by the recording
};

struct mirt_buf {
    mirt_ctx* ctx = nullptr;
    void* ptr = nullptr;
    size_t bytes = 0;
    bool owned = true;
    unsigned flags = 0;
    uint64_t uid = 0;      // unique per allocation: tells a live handle from a new object at a recycled address
    uint64_t version = 1;  // bumped by every host write / device zero / mirt_buf_invalidate
    // cell-offset validation cache
    uint64_t off_version = 0;
    uint32_t off_n = 0;
    uint32_t off_last = 0;
    uint32_t off_first = 0;       // off[0]: where the first cell's list starts (0 in every table a grid builder makes)
    uint32_t prep_count = 0;   // records in `prep` (the group spheres sit at record index prep_count)
    // prepared-triangle copy of a position buffer (fused path), rebuilt when the contents change
    void* prep = nullptr;
    size_t prep_bytes = 0;
    uint64_t prep_version = 0;
    uint64_t prep_gen = 1;    // bumped whenever `prep` is freed or replaced
    bool prep_sane = false;   // every plane-normal component is 0 or in [2^-40, 2^40]
};

static const std::vector<KernelSpec>& kernel_table() {
    // argument lists: A10 code.cl:440-1386 as bound by A10 code.js
    static const std::vector<KernelSpec> t = {
        {"sizeofRay", K_sizeofRay, {A_BUF}},
        {"sizeofPoi", K_sizeofPoi, {A_BUF}},
        {"initAcu", K_initAcu, {A_BUF, A_U32}},
        {"initTrace", K_initTrace, {A_BUF, A_BUF, A_BUF, A_AABB, A_F16, A_F32, A_F32, A_U32}},
        {"sphereTrace", K_sphereTrace, {A_U32, A_BUF, A_BUF, A_BUF, A_BUF, A_BUF, A_AABB, A_U32}},
        {"triangleTrace", K_triangleTrace, {A_U32, A_BUF, A_BUF, A_BUF, A_BUF, A_BUF, A_BUF, A_AABB, A_U32}},
        {"meshTrace", K_meshTrace, {A_U32, A_BUF, A_BUF, A_BUF, A_BUF, A_BUF, A_U32, A_AABB, A_U32}},
        {"lightRender", K_lightRender, {A_BUF, A_BUF, A_BUF, A_F16, A_U32}},
        {"initShadowTrace", K_initShadowTrace, {A_BUF, A_BUF, A_U32, A_F16, A_BUF}},
        {"sphereShadowTrace", K_sphereShadowTrace, {A_U32, A_BUF, A_BUF, A_BUF, A_AABB, A_U32}},
        {"triangleShadowTrace", K_triangleShadowTrace, {A_U32, A_BUF, A_BUF, A_BUF, A_AABB, A_U32}},
        {"sceneRender", K_sceneRender, {A_BUF, A_BUF, A_BUF, A_BUF, A_F16, A_U32}},
        {"bouncePaths", K_bouncePaths, {A_BUF, A_BUF, A_BUF, A_U32}},
        {"copyToPixel", K_copyToPixel, {A_BUF, A_BUF, A_F32, A_U32, A_U32}},
        // earlier assignments, selected with a dialect prefix (their kernel names collide with A10's):
        // A01 code.cl:116; A04 code.cl:200-315; A07 code.cl:307-626
        {"A01:raytrace", K_a01_raytrace, {A_BUF, A_F16}},
        {"A04:sizeofRay", K_a04_sizeofRay, {A_BUF}},
        {"A04:initTrace", K_a04_initTrace, {A_BUF, A_F16, A_BUF}},
        {"A04:meshTrace", K_a04_meshTrace, {A_BUF, A_F16, A_BUF, A_U32, A_BUF, A_BUF, A_BUF, A_BUF}},
        {"A07:sizeofRay", K_a07_sizeofRay, {A_BUF}},
        {"A07:initTrace", K_a07_initTrace, {A_BUF, A_F16, A_BUF, A_AABB}},
        {"A07:meshTrace", K_a07_meshTrace, {A_BUF, A_F16, A_BUF, A_U32, A_BUF, A_BUF, A_BUF, A_BUF, A_AABB, A_U32, A_BUF}},
        // molTrace(pixels, fcam, rays, s_size, s_atoms, s_mindex, m_color, bound, n_slabs, slab_size)   A07 code.cl:337-344
        {"A07:molTrace", K_a07_molTrace, {A_BUF, A_F16, A_BUF, A_U32, A_BUF, A_BUF, A_BUF, A_AABB, A_U32, A_BUF}},
    };
    return t;
}

struct mirt_kernel {
    mirt_ctx* ctx = nullptr;
    const KernelSpec* spec = nullptr;
    std::vector<KArg> args;
};

namespace {

bool live_has(const mirt_ctx* p) { return live_is(p, H_CTX); }
bool live_has(const mirt_buf* p) { return live_is(p, H_BUF); }
bool live_has(const mirt_kernel* p) { return live_is(p, H_KERNEL); }
bool live_has(const mirt_graph* p) { return live_is(p, H_GRAPH); }

int fail(mirt_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    t_last_error = buf;
    if (ctx && live_has(ctx)) ctx->last_error = buf;
    return code;
}

#define HIPCHK(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail((ctx), MIRT_E_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

size_t arg_size(ArgType t) {
    switch (t) {
        case A_U32: case A_F32: return 4;
        case A_F16: return 64;
        case A_AABB: return 32;
        default: return 0;
    }
}

uint32_t f2u_host(float f) {
    if (!(f == f)) return 0u;
    if (f >= 4294967296.0f) return UINT32_MAX;
    if (f <= 0.0f) return 0u;
    return (uint32_t)f;
}

// operations that wait on the stream or move host memory cannot be part of a recording
#define NOT_WHILE_CAPTURING(ctx, what) \
    do { if ((ctx)->capturing) return fail((ctx), MIRT_E_ARG, "%s is not possible inside mirt_capture_begin/end: run the sequence once before recording it", (what)); } while (0)

// A recording captures raw device pointers.  Every buffer a recorded launch touches is pinned: (handle, allocation uid), plus --
// `content` -- the version of its contents and the generation of its prepared copy when the launch relies on host-side
// validation or preparation of those contents (cell-offset tables, triangle positions).
void pin(mirt_ctx* ctx, const mirt_buf* b, bool content) {
    if (!ctx->capturing) return;
    for (auto& p : ctx->cap_pins)
        if (p.buf == b) { p.content = p.content || content; return; }
    ctx->cap_pins.push_back({const_cast<mirt_buf*>(b), b->uid, b->prep_gen, b->version, content});
}

int need(mirt_ctx* ctx, const char* what, const mirt_buf* b, uint64_t bytes) {
    if (!live_has(b)) return fail(ctx, MIRT_E_HANDLE, "%s: released or unknown buffer", what);
    if (b->ctx != ctx) return fail(ctx, MIRT_E_ARG, "%s: buffer belongs to another context", what);
    if ((uint64_t)b->bytes < bytes)
        return fail(ctx, MIRT_E_RANGE, "%s: buffer holds %zu bytes, launch needs %llu", what, b->bytes, (unsigned long long)bytes);
    pin(ctx, b, false);
    return MIRT_OK;
}

// Checks a cell-offset table (uint[n^3+1], non-decreasing) and that the primitive arrays hold
// off[n^3] entries.  The table is read back once per (buffer contents, n) and cached.
int check_grid(mirt_ctx* ctx, const char* what, mirt_buf* off, uint32_t n, const mirt_buf* prims, size_t prim_stride,
               const mirt_buf* normals, const mirt_buf* matid) {
    if (n == 0 || n > 1024) return fail(ctx, MIRT_E_ARG, "%s: n_slabs %u outside 1..1024", what, n);
    const uint64_t cells = (uint64_t)n * n * n;
    int rc = need(ctx, what, off, (cells + 1) * 4);
    if (rc) return rc;
    // memory the caller owns (mirt_buf_wrap) can change behind the runtime's back: its verdict is never cached
    if (!off->owned || !(off->off_version == off->version && off->off_n == n)) {
        NOT_WHILE_CAPTURING(ctx, "validating a new cell-offset table");
        std::vector<uint32_t> h(cells + 1);
        HIPCHK(ctx, hipMemcpyAsync(h.data(), off->ptr, (cells + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (uint64_t i = 0; i < cells; ++i) {
            if (h[i] > h[i + 1]) return fail(ctx, MIRT_E_DATA, "%s: cell offsets decrease at cell %llu", what, (unsigned long long)i);
            if (n > 1 && h[i + 1] - h[i] >= pt::kMaxCellSlots)
                return fail(ctx, MIRT_E_RANGE, "%s: cell %llu holds %u slots (a cell of a grid holds at most %u)", what, (unsigned long long)i, h[i + 1] - h[i], pt::kMaxCellSlots - 1u);
        }
        off->off_version = off->version;
        off->off_n = n;
        off->off_last = h[cells];
        off->off_first = h[0];
    }
    pin(ctx, off, true);
    const uint64_t count = off->off_last;
    rc = need(ctx, what, prims, count * prim_stride);
    if (rc) return rc;
    if (normals && (rc = need(ctx, what, normals, count * prim_stride))) return rc;
    if (matid && (rc = need(ctx, what, matid, count * 4))) return rc;
    return MIRT_OK;
}

int ensure_scratch(mirt_ctx* ctx, size_t bytes);

// n == 1 only.  A10 code.cl:699-707: x_next = pmin + (0 + (d>=0)) * ((pmax-pmin)/1).  When that reproduces pmax / pmin bit for
// bit on all three axes, the single cell's exit t is the very quotient interAABB already formed for the far slab plane.
uint32_t exit_is_far_face(const float* b8, uint32_t n) {
    if (n != 1) return 0;
    bool exact = true;
    for (int k = 0; k < 3; ++k) {
        volatile float lo = b8[k], hi = b8[4 + k];
        volatile float delta = (hi - lo) / 1.0f;
        volatile float up = lo + 1.0f * delta, dn = lo + 0.0f * delta;
        exact = exact && (up == hi) && (dn == lo) && (up == up);
    }
    return exact ? 1u : 0u;
}

// (re)builds the prepared-triangle copy of a position buffer when its contents changed
int ensure_prepared(mirt_ctx* ctx, mirt_buf* pb, uint32_t count) {
    // [count records of 48 B][one bounding sphere (float4) per group of records]: the spheres sit right behind the records, so a prepared copy is
    // good for exactly the count it was made for
    const size_t bytes = pt::prepared_bytes(count);
    if (pb->owned && pb->prep_version == pb->version && pb->prep_count == count && pb->prep_bytes >= bytes && (pb->prep || !bytes)) { pin(ctx, pb, true); return MIRT_OK; }
    NOT_WHILE_CAPTURING(ctx, "preparing a new triangle buffer");
    if (pb->prep && pb->prep_bytes < bytes) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(pb->prep)); pb->prep = nullptr; pb->prep_bytes = 0; pb->prep_gen++; }
    if (bytes && !pb->prep) { HIPCHK(ctx, hipMalloc(&pb->prep, bytes)); pb->prep_bytes = bytes; pb->prep_gen++; }
    int rc = ensure_scratch(ctx, 16);
    if (rc) return rc;
    uint32_t insane = 0;
    HIPCHK(ctx, hipMemsetAsync(ctx->scratch, 0, 4, ctx->stream));
    pt::launch_prepTriangles(ctx->stream, pb->ptr, pb->prep, count, (uint32_t*)ctx->scratch);
    HIPCHK(ctx, hipMemcpyAsync(&insane, ctx->scratch, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    pb->prep_sane = insane == 0;
    pb->prep_version = pb->version;
    if (pb->prep_count != count) pb->prep_gen++;   // rebuilt in place for another count: the group spheres and the sweep array moved, a recorded graph must not replay against it
    pb->prep_count = count;
    return MIRT_OK;
}

int ensure_scratch(mirt_ctx* ctx, size_t bytes) {
    if (ctx->capturing) ctx->cap_scratch = true;
    if (ctx->scratch_bytes >= bytes) return MIRT_OK;
    NOT_WHILE_CAPTURING(ctx, "growing the scratch buffer");
    if (ctx->scratch) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->scratch)); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
    ctx->scratch_gen++;
    HIPCHK(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return MIRT_OK;
}

// release paths shared by the public entry points and by mirt_ctx_destroy's reaping of leftovers
void free_buf(mirt_buf* buf, bool ctx_alive) {
    mirt_ctx* ctx = buf->ctx;
    live_del(buf);
    if (ctx_alive) {
        ctx->bufs.erase(buf);
        if ((buf->owned && buf->ptr) || buf->prep) {
            (void)hipSetDevice(ctx->device);
            (void)hipStreamSynchronize(ctx->stream);
        }
        if (buf->owned && buf->ptr) (void)hipFree(buf->ptr);
        if (buf->prep) (void)hipFree(buf->prep);
    }
    delete buf;
}
void free_kernel(mirt_kernel* k, bool ctx_alive) {
    live_del(k);
    if (ctx_alive) k->ctx->kernels.erase(k);
    delete k;
}
void free_graph(mirt_graph* g, bool ctx_alive) {
    live_del(g);
    if (ctx_alive) { g->ctx->graphs.erase(g); (void)hipStreamSynchronize(g->ctx->stream); }
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
}

}  // namespace

// No C++ exception crosses the C boundary (include/mirt.h): every entry point is a function-try-block.
namespace {
int caught(const char* fn, const char* what) noexcept {
    try { return fail(nullptr, MIRT_E_DEVICE, "%s: %s", fn, what); } catch (...) { return MIRT_E_DEVICE; }
}
}  // namespace
#define MIRT_CATCH(fn, ret)                                                         \
    catch (const std::bad_alloc&) { (void)caught(fn, "out of host memory"); ret; }  \
    catch (const std::exception& e) { (void)caught(fn, e.what()); ret; }            \
    catch (...) { (void)caught(fn, "unexpected C++ exception"); ret; }

extern "C" {

const char* mirt_version(void) try { return "mirt 0.2 (gfx950)"; } MIRT_CATCH("mirt_version", return "")

int mirt_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
} MIRT_CATCH("mirt_device_count", return MIRT_E_DEVICE)

int mirt_device_name(int device, char* out, size_t cap) try {
    if (!out || cap == 0) return fail(nullptr, MIRT_E_ARG, "mirt_device_name: null output");
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, MIRT_E_NODEVICE, "no HIP device %d", device); }
    snprintf(out, cap, "%s (%s, %d CUs)", p.name[0] ? p.name : "AMD Instinct (name not reported)", p.gcnArchName, p.multiProcessorCount);
    return MIRT_OK;
} MIRT_CATCH("mirt_device_name", return MIRT_E_DEVICE)

static int create_ctx(int device, mirt_ctx** out) {
    *out = nullptr;
    int n = mirt_device_count();
    if (n <= 0) return fail(nullptr, MIRT_E_NODEVICE, "no HIP device visible: libmirt needs an MI355X (gfx950); there is no CPU fallback");
    if (device < 0 || device >= n) return fail(nullptr, MIRT_E_ARG, "device %d out of range (0..%d)", device, n - 1);
    hipDeviceProp_t p;
    HIPCHK(nullptr, hipGetDeviceProperties(&p, device));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MIRT_E_NODEVICE, "device %d is %s; libmirt carries gfx950 code objects only", device, p.gcnArchName);
    HIPCHK(nullptr, hipSetDevice(device));
    mirt_ctx* c = new mirt_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(nullptr, MIRT_E_DEVICE, "hipStreamCreate failed"); }
    c->stream = c->own_stream;
    if (const char* f = getenv("MIRT_INPASS_RESOLVE")) c->inpass_resolve = atoi(f) != 0;   // A/B and test switch
    if (const char* f = getenv("MIRT_FUSION")) c->fusion = atoi(f) >= 2 ? 2 : 0;   // same as mirt_ctx_set_fusion: for hosts that cannot be edited at all
    (void)hipEventCreate(&c->ev0);
    (void)hipEventCreate(&c->ev1);
    for (auto& e : c->pe) (void)hipEventCreate(&e);
    live_add(c, H_CTX);
    *out = c;
    return MIRT_OK;
}

static int destroy_ctx(mirt_ctx* ctx) {
    (void)hipSetDevice(ctx->device);
    if (ctx->capturing) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(ctx->stream, &g); if (g) (void)hipGraphDestroy(g); ctx->capturing = false; }
    (void)flush_pending(ctx);   // held-back enqueues may write wrapped (caller-owned) memory: they still run
    (void)hipStreamSynchronize(ctx->stream);
    // whatever the host never released goes with the context; those handles become MIRT_E_HANDLE
    while (!ctx->graphs.empty()) free_graph(*ctx->graphs.begin(), true);
    while (!ctx->kernels.empty()) free_kernel(*ctx->kernels.begin(), true);
    while (!ctx->bufs.empty()) free_buf(*ctx->bufs.begin(), true);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->defer) (void)hipFree(ctx->defer);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (auto& e : ctx->pe) if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    live_del(ctx);
    delete ctx;
    return MIRT_OK;
}

int mirt_ctx_create(int device, mirt_ctx** out) try {
    if (!out) return fail(nullptr, MIRT_E_ARG, "mirt_ctx_create: null out");
    return create_ctx(device, out);
} MIRT_CATCH("mirt_ctx_create", return MIRT_E_DEVICE)

int mirt_ctx_destroy(mirt_ctx* ctx) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_ctx_destroy: unknown context");
    if (ctx->group) return fail(ctx, MIRT_E_ARG, "mirt_ctx_destroy: the context belongs to a device group; destroy the group");
    return destroy_ctx(ctx);
} MIRT_CATCH("mirt_ctx_destroy", return MIRT_E_DEVICE)

// ---- device groups: N contexts in one process + the one exchange of the path ---------------------------------------------
// RCCL is bound at run time (dlopen of librccl.so) the first time a group with more than one device -- or a forced RCCL gather --
// is created: a single-GPU host never loads it.
extern "C++" {
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
Rccl& rccl() {
    static Rccl r;
    if (r.lib || !r.err.empty()) return r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!r.lib) { r.err = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?"); return r; }
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p && r.err.empty()) r.err = std::string("librccl.so lacks ") + n; return p; };
    r.CommInitAll = (int (*)(void**, int, const int*))sym("ncclCommInitAll");
    r.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
    r.GroupStart = (int (*)())sym("ncclGroupStart");
    r.GroupEnd = (int (*)())sym("ncclGroupEnd");
    r.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
    r.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
    r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    return r;
}
constexpr int kNcclUint8 = 1;   // ncclUint8 / ncclChar family: rccl.h ncclDataType_t { ncclInt8 = 0, ncclUint8 = 1, ... }
}  // namespace
}  // extern "C++"

struct mirt_group {
    std::vector<mirt_ctx*> ctxs;
    std::vector<void*> comms;    // ncclComm_t per device; empty until a gather needs RCCL
    bool repeated_devices = false;   // rehearsal group (MIRT_GROUP_ALLOW_REPEATED_DEVICES): several contexts on one device
    std::vector<uint8_t> peer;   // [i * n + j]: context i's device reads context j's device's memory directly (hipDeviceEnablePeerAccess succeeded; 1 on the diagonal
                                 // and between contexts that share a device)
    std::vector<int> route;      // per tile, how the last mirt_gather moved it (MIRT_ROUTE_*)
};

static bool live_group(const mirt_group* g) { return live_is(g, H_GROUP); }

static int group_comms(mirt_group* g) {
    if (!g->comms.empty()) return MIRT_OK;
    Rccl& R = rccl();
    if (!R.err.empty()) return fail(g->ctxs[0], MIRT_E_DEVICE, "mirt_gather: %s", R.err.c_str());
    std::vector<int> devs;
    for (auto* c : g->ctxs) devs.push_back(c->device);
    g->comms.assign(devs.size(), nullptr);
    int rc = R.CommInitAll(g->comms.data(), (int)devs.size(), devs.data());
    if (rc != 0) { g->comms.clear(); return fail(g->ctxs[0], MIRT_E_DEVICE, "ncclCommInitAll over %zu devices: %s", devs.size(), R.GetErrorString(rc)); }
    return MIRT_OK;
}

int mirt_group_create(const int* device_ids, int n, mirt_group** out) try {
    if (!out) return fail(nullptr, MIRT_E_ARG, "mirt_group_create: null out");
    *out = nullptr;
    if (!device_ids || n < 1 || n > 64) return fail(nullptr, MIRT_E_ARG, "mirt_group_create: need 1..64 device ids");
    // A device may appear once -- unless MIRT_GROUP_ALLOW_REPEATED_DEVICES=1 (a rehearsal switch: N contexts share the devices at hand, so
    // the whole N-tile path -- tile arithmetic, N fused passes, the gather's offsets and orderings -- runs on a one-GPU box; such a group
    // gathers by device copies, RCCL wants one device per rank)
    bool repeated = false;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (device_ids[i] == device_ids[j]) repeated = true;
    if (repeated) {
        const char* e = getenv("MIRT_GROUP_ALLOW_REPEATED_DEVICES");
        if (!e || atoi(e) != 1) return fail(nullptr, MIRT_E_ARG, "mirt_group_create: a device is listed twice (MIRT_GROUP_ALLOW_REPEATED_DEVICES=1 allows it for rehearsals)");
    }
    mirt_group* g = new mirt_group();
    g->repeated_devices = repeated;
    for (int i = 0; i < n; ++i) {
        mirt_ctx* c = nullptr;
        int rc = create_ctx(device_ids[i], &c);
        if (rc) { for (auto* d : g->ctxs) { d->group = nullptr; destroy_ctx(d); } delete g; return rc; }
        c->group = g;
        g->ctxs.push_back(c);
    }
    // peer access both ways between every pair of distinct devices: what makes hipMemcpyPeerAsync (the copy transport of mirt_gather) a direct xGMI
    // transfer instead of a bounce through host memory.  A pair the runtime refuses stays 0 and its copies are reported as MIRT_ROUTE_STAGED.
    g->peer.assign((size_t)n * n, 0);
    g->route.assign((size_t)n, MIRT_ROUTE_NONE);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const int di = g->ctxs[i]->device, dj = g->ctxs[j]->device;
            if (di == dj) { g->peer[(size_t)i * n + j] = 1; continue; }
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, di, dj) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
            if (hipSetDevice(di) != hipSuccess) { (void)hipGetLastError(); continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) g->peer[(size_t)i * n + j] = 1;
            (void)hipGetLastError();
        }
    live_add(g, H_GROUP);
    *out = g;
    return MIRT_OK;
} MIRT_CATCH("mirt_group_create", return MIRT_E_DEVICE)

int mirt_group_peer_access(const mirt_group* g, int i, int j) try {
    if (!live_group(g)) return MIRT_E_HANDLE;
    const int n = (int)g->ctxs.size();
    if (i < 0 || j < 0 || i >= n || j >= n) return MIRT_E_ARG;
    return g->peer[(size_t)i * n + j];
} MIRT_CATCH("mirt_group_peer_access", return MIRT_E_DEVICE)

int mirt_gather_route(const mirt_group* g, int tile) try {
    if (!live_group(g)) return MIRT_E_HANDLE;
    if (tile < 0 || tile >= (int)g->route.size()) return MIRT_E_ARG;
    return g->route[(size_t)tile];
} MIRT_CATCH("mirt_gather_route", return MIRT_E_DEVICE)

int mirt_abi_version(void) { return MIRT_ABI_VERSION; }

int mirt_group_size(const mirt_group* g) try { return live_group(g) ? (int)g->ctxs.size() : MIRT_E_HANDLE; } MIRT_CATCH("mirt_group_size", return MIRT_E_DEVICE)

mirt_ctx* mirt_group_ctx(const mirt_group* g, int index) try {
    if (!live_group(g) || index < 0 || index >= (int)g->ctxs.size()) { fail(nullptr, MIRT_E_ARG, "mirt_group_ctx: unknown group or index out of range"); return nullptr; }
    return g->ctxs[index];
} MIRT_CATCH("mirt_group_ctx", return nullptr)

int mirt_group_destroy(mirt_group* g) try {
    if (!live_group(g)) return fail(nullptr, MIRT_E_HANDLE, "mirt_group_destroy: unknown group");
    for (auto* c : g->ctxs) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
    if (!g->comms.empty()) for (void* c : g->comms) if (c) (void)rccl().CommDestroy(c);
    for (size_t i = 0; i < g->ctxs.size(); ++i) {
        g->ctxs[i]->group = nullptr;
        destroy_ctx(g->ctxs[i]);
    }
    live_del(g);
    delete g;
    return MIRT_OK;
} MIRT_CATCH("mirt_group_destroy", return MIRT_E_DEVICE)

void mirt_tile_rows(uint32_t height, uint32_t n_tiles, uint32_t index, uint32_t* row0, uint32_t* nrows) try {
    // contiguous row tiles whose sizes differ by at most one row: the first height % n tiles take the extra row
    if (!n_tiles || index >= n_tiles) { if (row0) *row0 = 0; if (nrows) *nrows = 0; return; }
    const uint32_t q = height / n_tiles, r = height % n_tiles;
    if (row0) *row0 = index * q + (index < r ? index : r);
    if (nrows) *nrows = q + (index < r ? 1u : 0u);
} MIRT_CATCH("mirt_tile_rows", return)

int mirt_gather(mirt_group* g, mirt_buf* const* tiles, const size_t* tile_bytes, int n_tiles, mirt_buf* out, int root, int transport) try {
    if (!live_group(g)) return fail(nullptr, MIRT_E_HANDLE, "mirt_gather: unknown group");
    const int n = (int)g->ctxs.size();
    mirt_ctx* rc_ctx = g->ctxs[0];
    // arguments first: nothing is flushed or queued on any device for a call that is going to be refused
    if (!tiles || !tile_bytes || root < 0 || root >= n) return fail(rc_ctx, MIRT_E_ARG, "mirt_gather: null argument or root out of range");
    if (n_tiles != n) return fail(rc_ctx, MIRT_E_ARG, "mirt_gather: %d tiles for a group of %d contexts", n_tiles, n);
    if (transport != MIRT_GATHER_AUTO && transport != MIRT_GATHER_RCCL && transport != MIRT_GATHER_COPY)
        return fail(rc_ctx, MIRT_E_ARG, "mirt_gather: transport %d is not MIRT_GATHER_AUTO / _RCCL / _COPY", transport);
    mirt_ctx* rootc = g->ctxs[root];
    uint64_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (!live_has(tiles[i]) || tiles[i]->ctx != g->ctxs[i]) return fail(rc_ctx, MIRT_E_HANDLE, "mirt_gather: tile %d is not a live buffer of the group's context %d", i, i);
        if (tiles[i]->bytes < tile_bytes[i]) return fail(rc_ctx, MIRT_E_RANGE, "mirt_gather: tile %d holds %zu bytes, asked to send %zu", i, tiles[i]->bytes, tile_bytes[i]);
        total += tile_bytes[i];
    }
    if (!live_has(out) || out->ctx != rootc) return fail(rc_ctx, MIRT_E_HANDLE, "mirt_gather: the output is not a live buffer of the root context");
    if (out->bytes < total) return fail(rc_ctx, MIRT_E_RANGE, "mirt_gather: output holds %zu bytes, the tiles add up to %llu", out->bytes, (unsigned long long)total);
    for (auto* c : g->ctxs) NOT_WHILE_CAPTURING(c, "mirt_gather");
    for (auto* c : g->ctxs) FLUSH_PENDING(c);
    // MIRT_GATHER_AUTO: RCCL when the group spans several distinct devices and librccl loads, else copies
    bool use_rccl = transport == MIRT_GATHER_RCCL;
    if (transport == MIRT_GATHER_AUTO && n > 1 && !g->repeated_devices) use_rccl = rccl().err.empty();
    if (use_rccl && g->repeated_devices) return fail(rc_ctx, MIRT_E_ARG, "mirt_gather: RCCL needs one device per context (this group lists a device twice: MIRT_GATHER_COPY)");
    if (!use_rccl) {
        // copies queued on the root's stream, each ordered after the work queued so far on the tile's own stream: tile == frame for one
        // context; device-to-device across contexts that share a device; hipMemcpyPeerAsync (xGMI, P2P) across devices
        HIPCHK(rootc, hipSetDevice(rootc->device));
        uint64_t off = 0;
        for (int i = 0; i < n; ++i) {
            mirt_ctx* c = g->ctxs[i];
            if (tile_bytes[i]) {
                if (c->stream != rootc->stream) {
                    hipEvent_t ev;
                    HIPCHK(rootc, hipSetDevice(c->device));
                    HIPCHK(rootc, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                    hipError_t e = hipEventRecord(ev, c->stream);
                    if (e == hipSuccess) { (void)hipSetDevice(rootc->device); e = hipStreamWaitEvent(rootc->stream, ev, 0); }
                    (void)hipEventDestroy(ev);   // released once the wait has been satisfied
                    HIPCHK(rootc, hipSetDevice(rootc->device));
                    HIPCHK(rootc, e);
                }
                if (c->device == rootc->device) HIPCHK(rootc, hipMemcpyAsync((char*)out->ptr + off, tiles[i]->ptr, tile_bytes[i], hipMemcpyDeviceToDevice, rootc->stream));
                else HIPCHK(rootc, hipMemcpyPeerAsync((char*)out->ptr + off, rootc->device, tiles[i]->ptr, c->device, tile_bytes[i], rootc->stream));
                g->route[(size_t)i] = c->device == rootc->device ? MIRT_ROUTE_LOCAL : (g->peer[(size_t)root * n + i] ? MIRT_ROUTE_PEER : MIRT_ROUTE_STAGED);
            } else g->route[(size_t)i] = MIRT_ROUTE_NONE;
            off += tile_bytes[i];
        }
        out->version++;
        return MIRT_OK;
    }
    int rc = group_comms(g);
    if (rc) return rc;
    Rccl& R = rccl();
    // one grouped exchange: every device sends its tile to the root (the root's own tile included: a self send/recv pair is legal
    // inside a group), the root receives each at that tile's offset.  N - 1 peers -> N - 1 distinct xGMI links into the root.
    int nrc = R.GroupStart();
    uint64_t off = 0;
    for (int i = 0; i < n && nrc == 0; ++i) {
        if (tile_bytes[i]) {
            (void)hipSetDevice(g->ctxs[i]->device);
            nrc = R.Send(tiles[i]->ptr, tile_bytes[i], kNcclUint8, root, g->comms[i], g->ctxs[i]->stream);
            (void)hipSetDevice(rootc->device);
            if (nrc == 0) nrc = R.Recv((char*)out->ptr + off, tile_bytes[i], kNcclUint8, i, g->comms[root], rootc->stream);
        }
        off += tile_bytes[i];
    }
    const int erc = R.GroupEnd();
    (void)hipSetDevice(rootc->device);
    if (nrc == 0) nrc = erc;
    if (nrc != 0) return fail(rc_ctx, MIRT_E_DEVICE, "mirt_gather: RCCL: %s", R.GetErrorString(nrc));
    for (int i = 0; i < n; ++i) g->route[(size_t)i] = tile_bytes[i] ? MIRT_ROUTE_RCCL : MIRT_ROUTE_NONE;
    out->version++;
    return MIRT_OK;
} MIRT_CATCH("mirt_gather", return MIRT_E_DEVICE)

int mirt_group_finish(mirt_group* g) try {
    if (!live_group(g)) return fail(nullptr, MIRT_E_HANDLE, "mirt_group_finish: unknown group");
    for (auto* c : g->ctxs) {
        NOT_WHILE_CAPTURING(c, "mirt_group_finish");
        FLUSH_PENDING(c);
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MIRT_OK;
} MIRT_CATCH("mirt_group_finish", return MIRT_E_DEVICE)

const char* mirt_last_error(mirt_ctx* ctx) try {
    if (ctx && live_has(ctx)) return ctx->last_error.c_str();
    return t_last_error.c_str();
} MIRT_CATCH("mirt_last_error", return "")

int mirt_ctx_set_stream(mirt_ctx* ctx, void* hip_stream) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_ctx_set_stream: unknown context");
    NOT_WHILE_CAPTURING(ctx, "mirt_ctx_set_stream");
    FLUSH_PENDING(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return MIRT_OK;
} MIRT_CATCH("mirt_ctx_set_stream", return MIRT_E_DEVICE)

int mirt_finish(mirt_ctx* ctx) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_finish: unknown context");
    NOT_WHILE_CAPTURING(ctx, "mirt_finish");
    // inside a held pass (the reference calls finish() after every sceneRender, A10 code.js:1406) nothing is observable until a read, which
    // flushes -- unless the pass works on memory the caller can reach behind the ABI (mirt_buf_wrap): then finish() means finish
    if (ctx->fusion >= 2 && !ctx->pending.empty()) {
        bool wrapped = false;
        for (const auto& p : ctx->pending)
            for (size_t j = 0; j < p.args.size(); ++j)
                if (p.spec->args[j] == A_BUF && live_has(p.args[j].buf) && !p.args[j].buf->owned) wrapped = true;
        if (!wrapped) return MIRT_OK;
        FLUSH_PENDING(ctx);
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MIRT_OK;
} MIRT_CATCH("mirt_finish", return MIRT_E_DEVICE)

int mirt_buf_create(mirt_ctx* ctx, size_t bytes, unsigned flags, mirt_buf** out) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_buf_create: unknown context");
    if (!out) return fail(ctx, MIRT_E_ARG, "mirt_buf_create: null out");
    *out = nullptr;
    if (bytes == 0) return fail(ctx, MIRT_E_ARG, "mirt_buf_create: zero-size buffer (WebCL INVALID_BUFFER_SIZE)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    HIPCHK(ctx, hipMalloc(&p, bytes));
    mirt_buf* b = new mirt_buf();
    b->ctx = ctx; b->ptr = p; b->bytes = bytes; b->owned = true; b->flags = flags; b->uid = g_next_uid++;
    live_add(b, H_BUF);
    ctx->bufs.insert(b);
    *out = b;
    return MIRT_OK;
} MIRT_CATCH("mirt_buf_create", return MIRT_E_DEVICE)

int mirt_buf_wrap(mirt_ctx* ctx, void* device_ptr, size_t bytes, mirt_buf** out) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_buf_wrap: unknown context");
    if (!out || !device_ptr || bytes == 0) return fail(ctx, MIRT_E_ARG, "mirt_buf_wrap: null pointer or zero size");
    if (((uintptr_t)device_ptr & 15u) != 0) return fail(ctx, MIRT_E_ARG, "mirt_buf_wrap: device pointer must be 16-byte aligned");
    mirt_buf* b = new mirt_buf();
    b->ctx = ctx; b->ptr = device_ptr; b->bytes = bytes; b->owned = false; b->flags = MIRT_MEM_READ_WRITE; b->uid = g_next_uid++;
    live_add(b, H_BUF);
    ctx->bufs.insert(b);
    *out = b;
    return MIRT_OK;
} MIRT_CATCH("mirt_buf_wrap", return MIRT_E_DEVICE)

int mirt_buf_release(mirt_buf* buf) try {
    if (!live_has(buf)) return fail(nullptr, MIRT_E_HANDLE, "mirt_buf_release: unknown or already released buffer");
    if (live_has(buf->ctx)) { mirt_ctx* ctx = buf->ctx; FLUSH_PENDING(ctx); }
    free_buf(buf, live_has(buf->ctx));
    return MIRT_OK;
} MIRT_CATCH("mirt_buf_release", return MIRT_E_DEVICE)

int mirt_buf_invalidate(mirt_buf* buf) try {
    if (!live_has(buf)) return fail(nullptr, MIRT_E_HANDLE, "mirt_buf_invalidate: unknown buffer");
    buf->version++;
    return MIRT_OK;
} MIRT_CATCH("mirt_buf_invalidate", return MIRT_E_DEVICE)

size_t mirt_buf_size(const mirt_buf* buf) try { return live_has(buf) ? buf->bytes : 0; } MIRT_CATCH("mirt_buf_size", return 0)
void* mirt_buf_device_ptr(const mirt_buf* buf) try {
    if (!live_has(buf)) return nullptr;
    if (live_has(buf->ctx)) (void)flush_pending(buf->ctx);   // the caller is about to look at the memory itself
    return buf->ptr;
} MIRT_CATCH("mirt_buf_device_ptr", return nullptr)

int mirt_buf_write(mirt_buf* buf, size_t offset, size_t nbytes, const void* host, int blocking) try {
    (void)blocking;
    if (!live_has(buf)) return fail(nullptr, MIRT_E_HANDLE, "mirt_buf_write: unknown buffer");
    mirt_ctx* ctx = buf->ctx;
    NOT_WHILE_CAPTURING(ctx, "mirt_buf_write");
    FLUSH_PENDING(ctx);
    if (!host && nbytes) return fail(ctx, MIRT_E_ARG, "mirt_buf_write: null host pointer");
    if (offset > buf->bytes || nbytes > buf->bytes - offset)
        return fail(ctx, MIRT_E_RANGE, "mirt_buf_write: [%zu, +%zu) exceeds buffer of %zu bytes", offset, nbytes, buf->bytes);
    if (!nbytes) return MIRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync((char*)buf->ptr + offset, host, nbytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the host array is borrowed for this call only
    buf->version++;
    return MIRT_OK;
} MIRT_CATCH("mirt_buf_write", return MIRT_E_DEVICE)

int mirt_buf_read(mirt_buf* buf, size_t offset, size_t nbytes, void* host, int blocking) try {
    (void)blocking;
    if (!live_has(buf)) return fail(nullptr, MIRT_E_HANDLE, "mirt_buf_read: unknown buffer");
    mirt_ctx* ctx = buf->ctx;
    NOT_WHILE_CAPTURING(ctx, "mirt_buf_read");
    FLUSH_PENDING(ctx);
    if (!host && nbytes) return fail(ctx, MIRT_E_ARG, "mirt_buf_read: null host pointer");
    if (offset > buf->bytes || nbytes > buf->bytes - offset)
        return fail(ctx, MIRT_E_RANGE, "mirt_buf_read: [%zu, +%zu) exceeds buffer of %zu bytes", offset, nbytes, buf->bytes);
    if (!nbytes) return MIRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(host, (const char*)buf->ptr + offset, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MIRT_OK;
} MIRT_CATCH("mirt_buf_read", return MIRT_E_DEVICE)

// OpenCL C text with comments blanked out, and the names of its `__kernel void NAME(` definitions
static void scan_source(const char* source, std::string* code, std::vector<std::string>* kernels) {
    const char* p = source;
    while (*p) {
        if (p[0] == '/' && p[1] == '*') { p += 2; while (*p && !(p[0] == '*' && p[1] == '/')) ++p; if (*p) p += 2; code->push_back(' '); continue; }
        if (p[0] == '/' && p[1] == '/') { while (*p && *p != '\n') ++p; continue; }
        code->push_back(*p++);
    }
    size_t pos = 0;
    while ((pos = code->find("__kernel", pos)) != std::string::npos) {
        size_t q = pos + 8;
        auto ws = [&](size_t i) { while (i < code->size() && strchr(" \t\r\n", (*code)[i])) ++i; return i; };
        q = ws(q);
        if (code->compare(q, 4, "void") == 0) {
            q = ws(q + 4);
            size_t e = q;
            while (e < code->size() && (isalnum((unsigned char)(*code)[e]) || (*code)[e] == '_')) ++e;
            if (e > q) kernels->push_back(code->substr(q, e - q));
        }
        pos += 8;
    }
}

int mirt_program_dialect(const char* source) try {
    if (!source) return 0;
    std::string code;
    std::vector<std::string> ks;
    scan_source(source, &code, &ks);
    auto has = [&](const char* k) { for (auto& n : ks) if (n == k) return true; return false; };
    auto text = [&](const char* t) { return code.find(t) != std::string::npos; };
    if (has("bouncePaths") && has("sceneRender")) return 10;
    if (has("sceneRender") || has("initShadowTrace")) return 0;      // A08 / A09: earlier drafts of the A10 set, not built
    if (has("meshTrace")) {
        if (text("z_stride")) return 7;                              // 3-D grid (A07 code.cl:545)
        if (!text("n_slabs") && !text("AABB bound")) return 4;       // brute force, no pre-clip (A04)
        return 0;                                                    // A05 / A06: not built
    }
    if (ks.size() == 1 && ks[0] == "raytrace" && !text("__global float4")) return 1;   // A02's raytrace takes atom arrays
    return 0;
} MIRT_CATCH("mirt_program_dialect", return MIRT_E_DEVICE)

int mirt_program_check(mirt_ctx* ctx, const char* source, char* missing, size_t cap) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_program_check: unknown context");
    if (!source) return fail(ctx, MIRT_E_ARG, "mirt_program_check: null source");
    std::string code, miss;
    std::vector<std::string> ks;
    scan_source(source, &code, &ks);
    const int dialect = mirt_program_dialect(source);
    const std::string prefix = dialect == 7 ? "A07:" : dialect == 4 ? "A04:" : dialect == 1 ? "A01:" : "";
    int n_missing = 0;
    for (const auto& name : ks) {
        bool found = false;
        for (const auto& k : kernel_table()) if (prefix + name == k.name) found = true;
        if (!found) { if (n_missing++) miss += ","; miss += name; }
    }
    if (missing && cap) snprintf(missing, cap, "%s", miss.c_str());
    return n_missing;
} MIRT_CATCH("mirt_program_check", return MIRT_E_DEVICE)

int mirt_kernel_get(mirt_ctx* ctx, const char* name, mirt_kernel** out) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_kernel_get: unknown context");
    if (!name || !out) return fail(ctx, MIRT_E_ARG, "mirt_kernel_get: null argument");
    *out = nullptr;
    for (const auto& s : kernel_table()) {
        if (strcmp(name, s.name) == 0) {
            mirt_kernel* k = new mirt_kernel();
            k->ctx = ctx; k->spec = &s; k->args.resize(s.args.size());
            live_add(k, H_KERNEL);
            ctx->kernels.insert(k);
            *out = k;
            return MIRT_OK;
        }
    }
    return fail(ctx, MIRT_E_NAME, "no kernel named '%s' (INVALID_KERNEL_NAME)", name);
} MIRT_CATCH("mirt_kernel_get", return MIRT_E_DEVICE)

int mirt_kernel_release(mirt_kernel* k) try {
    if (!live_has(k)) return fail(nullptr, MIRT_E_HANDLE, "mirt_kernel_release: unknown or already released kernel");
    free_kernel(k, live_has(k->ctx));
    return MIRT_OK;
} MIRT_CATCH("mirt_kernel_release", return MIRT_E_DEVICE)

int mirt_kernel_num_args(const mirt_kernel* k) try { return live_has(k) ? (int)k->spec->args.size() : MIRT_E_HANDLE; } MIRT_CATCH("mirt_kernel_num_args", return MIRT_E_DEVICE)
int mirt_kernel_preferred_multiple(const mirt_kernel* k) try { return live_has(k) ? 64 : MIRT_E_HANDLE; } MIRT_CATCH("mirt_kernel_preferred_multiple", return MIRT_E_DEVICE)

int mirt_kernel_set_arg(mirt_kernel* k, unsigned index, size_t size, const void* value) try {
    if (!live_has(k)) return fail(nullptr, MIRT_E_HANDLE, "mirt_kernel_set_arg: unknown kernel");
    mirt_ctx* ctx = k->ctx;
    if (index >= k->args.size()) return fail(ctx, MIRT_E_ARG, "%s: argument index %u out of range (%zu args)", k->spec->name, index, k->args.size());
    ArgType t = k->spec->args[index];
    if (t == A_BUF) return fail(ctx, MIRT_E_ARG, "%s: argument %u is a buffer; use mirt_kernel_set_arg_buf", k->spec->name, index);
    if (!value || size != arg_size(t))
        return fail(ctx, MIRT_E_ARG, "%s: argument %u expects %zu bytes, got %zu (INVALID_ARG_SIZE)", k->spec->name, index, arg_size(t), size);
    memcpy(&k->args[index].val, value, size);
    k->args[index].set = true;
    return MIRT_OK;
} MIRT_CATCH("mirt_kernel_set_arg", return MIRT_E_DEVICE)

int mirt_kernel_set_arg_buf(mirt_kernel* k, unsigned index, mirt_buf* buf) try {
    if (!live_has(k)) return fail(nullptr, MIRT_E_HANDLE, "mirt_kernel_set_arg_buf: unknown kernel");
    mirt_ctx* ctx = k->ctx;
    if (index >= k->args.size()) return fail(ctx, MIRT_E_ARG, "%s: argument index %u out of range", k->spec->name, index);
    if (k->spec->args[index] != A_BUF) return fail(ctx, MIRT_E_ARG, "%s: argument %u is not a buffer", k->spec->name, index);
    if (!live_has(buf)) return fail(ctx, MIRT_E_HANDLE, "%s: argument %u: unknown or released buffer", k->spec->name, index);
    if (buf->ctx != ctx) return fail(ctx, MIRT_E_ARG, "%s: argument %u: buffer belongs to another context", k->spec->name, index);
    k->args[index].buf = buf;
    k->args[index].set = true;
    return MIRT_OK;
} MIRT_CATCH("mirt_kernel_set_arg_buf", return MIRT_E_DEVICE)

// ---- command-stream fusion ---------------------------------------------------------------------------------------------------
// The reference host issues one pass as 44+ enqueues (A10 code.js:1806-1854), every stage round-tripping Ray / Poi / shadow Ray / acu
// through HBM: 4.2 KB per sample against the fused pass's 24 B.  With mirt_ctx_set_fusion(ctx, 2) the runtime holds the enqueues of
// the Assign10 pass kernels back, and when the stream from an initTrace up to a copyToPixel IS executeRender's sequence over one
// consistent set of buffers -- initTrace, the closest-hit kernels, lightRender per light, per light {initShadowTrace, any-hit kernels,
// sceneRender}, any number of {bouncePaths, closest-hit kernels, per-light block}, copyToPixel -- it runs the pass as ONE launch of
// k_fusedPass (render_pass_impl) + the recorded copyToPixel.  Anything else (a different order, mixed buffers, a read / write / release /
// other command in between) flushes the held enqueues one by one, unchanged.  See include/mirt.h for what the mode trades.
static int render_pass_impl(mirt_ctx* ctx, const mirt_pass_desc* d, bool fresh);
static int launch_kernel(mirt_ctx* ctx, const KernelSpec& S, std::vector<KArg>& a, unsigned dim, const size_t* global);

static bool same_f(const float* x, const float* y, int n) { return memcmp(x, y, (size_t)n * 4) == 0; }

// MIRT_OK: the held stream was a whole pass and has been executed fused.  1: not a pass (nothing executed).  < 0: error.
static int try_fuse_pass(mirt_ctx* ctx, std::vector<mirt_ctx::Pending>& P) {
    size_t i = 0;
    auto id = [&](size_t k) { return k < P.size() ? P[k].spec->id : K_COUNT; };
#define PB(k, j) (P[k].args[j].buf)
#define PU(k, j) (P[k].args[j].val.u)
#define PF(k, j) (P[k].args[j].val.f)
#define PV(k, j) (P[k].args[j].val.v)
    if (id(0) != K_initTrace || P[0].dim != 2) return 1;
    mirt_buf *seeds = PB(0, 0), *rays = PB(0, 1), *pois = PB(0, 2);
    const uint32_t rpp = PU(0, 7);
    const uint32_t cols = f2u_host(PV(0, 4)[14]), rows = f2u_host(PV(0, 4)[15]);
    if (!rpp || !cols || !rows || P[0].g[0] < cols || P[0].g[1] < rows) return 1;
    const uint64_t total64 = (uint64_t)cols * rows * rpp;
    if (total64 > 0xFFFFFFFFull) return 1;
    const uint32_t total = (uint32_t)total64;
    i = 1;
    struct SetRef { int kind; mirt_buf *prims, *normals, *matid, *off; float bounds[8]; uint32_t n, mesh_matid; };
    auto closest_group = [&](std::vector<SetRef>& out) -> bool {
        for (;; ++i) {
            const KernelId k = id(i);
            if (k != K_sphereTrace && k != K_triangleTrace && k != K_meshTrace) return true;
            if (PU(i, 0) != total || PB(i, 1) != pois || PB(i, 2) != rays || P[i].g[0] < total) return false;
            SetRef r;
            memset(&r, 0, sizeof r);
            if (k == K_sphereTrace) { r.kind = 0; r.prims = PB(i, 3); r.matid = PB(i, 4); r.off = PB(i, 5); memcpy(r.bounds, PV(i, 6), 32); r.n = PU(i, 7); }
            else if (k == K_triangleTrace) { r.kind = 1; r.prims = PB(i, 3); r.normals = PB(i, 4); r.matid = PB(i, 5); r.off = PB(i, 6); memcpy(r.bounds, PV(i, 7), 32); r.n = PU(i, 8); }
            else { r.kind = 2; r.prims = PB(i, 3); r.normals = PB(i, 4); r.off = PB(i, 5); r.mesh_matid = PU(i, 6); memcpy(r.bounds, PV(i, 7), 32); r.n = PU(i, 8); }
            out.push_back(r);
        }
    };
    std::vector<SetRef> sets;
    if (!closest_group(sets)) return 1;
    // upload order the fused pass assumes: at most one sphere set, then at most one loose-triangle set, then the meshes
    {
        size_t k = 0;
        if (k < sets.size() && sets[k].kind == 0) ++k;
        if (k < sets.size() && sets[k].kind == 1) ++k;
        for (; k < sets.size(); ++k) if (sets[k].kind != 2) return 1;
        if (sets.size() > 2u + MIRT_MAX_MESHES) return 1;
    }
    struct LightRef { float light[16], shadow[16], scene[16]; };
    std::vector<LightRef> lights;
    mirt_buf *acu = nullptr, *shadow = nullptr, *material = nullptr;
    for (; id(i) == K_lightRender; ++i) {
        if (PB(i, 0) != pois || PB(i, 1) != rays || PU(i, 4) != total || P[i].g[0] < total) return 1;
        if (acu && PB(i, 2) != acu) return 1;
        acu = PB(i, 2);
        LightRef L;
        memset(&L, 0, sizeof L);
        memcpy(L.light, PV(i, 3), 64);
        lights.push_back(L);
    }
    if (lights.size() > MIRT_MAX_LIGHTS) return 1;
    auto direct_block = [&](bool first) -> bool {
        for (size_t l = 0; l < lights.size(); ++l) {
            if (id(i) != K_initShadowTrace || PB(i, 1) != pois || PU(i, 2) != total || PB(i, 4) != seeds || P[i].g[0] < total) return false;
            if (shadow && PB(i, 0) != shadow) return false;
            shadow = PB(i, 0);
            if (first) memcpy(lights[l].shadow, PV(i, 3), 64); else if (!same_f(lights[l].shadow, PV(i, 3), 16)) return false;
            ++i;
            for (const SetRef& r : sets) {   // one any-hit kernel per set, same order, same geometry
                const KernelId want = r.kind == 0 ? K_sphereShadowTrace : K_triangleShadowTrace;
                if (id(i) != want || PU(i, 0) != total || PB(i, 1) != shadow || PB(i, 2) != r.prims || PB(i, 3) != r.off ||
                    !same_f(PV(i, 4), r.bounds, 8) || PU(i, 5) != r.n || P[i].g[0] < total) return false;
                ++i;
            }
            if (id(i) != K_sceneRender || PB(i, 0) != acu || PB(i, 1) != pois || PB(i, 2) != shadow || PU(i, 5) != total || P[i].g[0] < total) return false;
            if (material && PB(i, 3) != material) return false;
            material = PB(i, 3);
            if (first) memcpy(lights[l].scene, PV(i, 4), 64); else if (!same_f(lights[l].scene, PV(i, 4), 16)) return false;
            ++i;
        }
        return true;
    };
    if (lights.empty() || !direct_block(true)) return 1;     // a scene without lights has no sceneRender to take acu / material from: not fused
    uint32_t bounces = 0;
    while (id(i) == K_bouncePaths) {
        if (PB(i, 0) != pois || PB(i, 1) != rays || PB(i, 2) != seeds || PU(i, 3) != total || P[i].g[0] < total) return 1;
        ++i;
        std::vector<SetRef> again;
        if (!closest_group(again) || again.size() != sets.size()) return 1;
        for (size_t k = 0; k < sets.size(); ++k)
            if (again[k].kind != sets[k].kind || again[k].prims != sets[k].prims || again[k].normals != sets[k].normals || again[k].matid != sets[k].matid ||
                again[k].off != sets[k].off || again[k].n != sets[k].n || again[k].mesh_matid != sets[k].mesh_matid || !same_f(again[k].bounds, sets[k].bounds, 8)) return 1;
        if (!direct_block(false)) return 1;
        ++bounces;
    }
    if (id(i) != K_copyToPixel || i + 1 != P.size()) return 1;
    if (PB(i, 1) != acu || PU(i, 3) != cols * rows || PU(i, 4) != rpp || P[i].g[0] < cols * rows) return 1;
    {   // the fused pass takes k x k rays per pixel only (see render_pass_impl)
        const uint32_t k = (uint32_t)std::sqrt((double)rpp);
        const uint32_t kk = (k + 1) * (k + 1) == rpp ? k + 1 : k;
        if (kk * kk != rpp) return 1;
    }
    // ---- it is a pass: one fused launch + the recorded copyToPixel
    mirt_pass_desc d;
    memset(&d, 0, sizeof d);
    d.struct_size = sizeof d;
    d.width = cols; d.height = rows; d.rays_per_pixel = rpp; d.bounces = bounces; d.pass_index = 1;
    memcpy(d.cam, PV(0, 4), 64);
    memcpy(d.scene_bounds, PV(0, 3), 32);
    d.focal_length = PF(0, 5); d.lens_rad = PF(0, 6);
    std::vector<mirt_grid> grids(sets.size());
    std::vector<mirt_light> ls(lights.size());
    for (size_t k = 0; k < sets.size(); ++k) {
        grids[k].prims = sets[k].prims; grids[k].normals = sets[k].normals; grids[k].matid = sets[k].matid; grids[k].cell_offsets = sets[k].off;
        memcpy(grids[k].bounds, sets[k].bounds, 32);
        grids[k].n_slabs = sets[k].n; grids[k].mesh_matid = sets[k].mesh_matid;
    }
    size_t k = 0;
    if (k < sets.size() && sets[k].kind == 0) d.spheres = &grids[k++];
    if (k < sets.size() && sets[k].kind == 1) d.triangles = &grids[k++];
    d.meshes = k < sets.size() ? &grids[k] : nullptr;
    d.n_meshes = (uint32_t)(sets.size() - k);
    for (size_t l = 0; l < lights.size(); ++l) { memcpy(ls[l].light, lights[l].light, 64); memcpy(ls[l].shadow, lights[l].shadow, 64); memcpy(ls[l].scene, lights[l].scene, 64); }
    d.lights = ls.data(); d.n_lights = (uint32_t)ls.size();
    d.material = material; d.seeds = seeds; d.acu = acu;
    // the recorded copyToPixel goes into the pass where the pass can resolve its own pixels (whole pixels per block of 256 ray ids): its pixel buffer and
    // its factor as the host passed it; where it cannot, render_pass_impl queues the separate kernel behind the pass, with that factor
    d.pixel = PB(i, 0);
    ctx->res_m_override = PF(i, 2);
    int rc = render_pass_impl(ctx, &d, false);
    ctx->res_m_override = NAN;
    if (rc && rc != MIRT_E_DEVICE) return 1;   // refused before anything was launched (a size, a grid that fails validation ...): run the stream as issued,
    if (rc) return rc;                          // whose own checks then report it against the kernel that trips it
    PB(i, 0)->version++;   // (written by the pass itself, or by the copyToPixel render_pass_impl queued behind it with the recorded factor)
    ctx->fused_passes++;
    return MIRT_OK;
#undef PB
#undef PU
#undef PF
#undef PV
}

// run whatever is held back: as one fused pass when it is one, else enqueue by enqueue, unchanged
static int flush_pending(mirt_ctx* ctx) {
    if (ctx->pending.empty()) return MIRT_OK;
    std::vector<mirt_ctx::Pending> P;
    P.swap(ctx->pending);
    for (auto& p : P)
        for (size_t j = 0; j < p.args.size(); ++j)
            if (p.spec->args[j] == A_BUF && (!live_has(p.args[j].buf) || p.args[j].buf->ctx != ctx))
                return fail(ctx, MIRT_E_HANDLE, "%s: a buffer of a held-back enqueue was released", p.spec->name);
    int rc = try_fuse_pass(ctx, P);
    if (rc <= 0) return rc;
    for (auto& p : P)
        if ((rc = launch_kernel(ctx, *p.spec, p.args, p.dim, p.g))) return rc;
    return MIRT_OK;
}

static bool is_pass_kernel(KernelId k) { return k >= K_initTrace && k <= K_copyToPixel; }

int mirt_enqueue(mirt_ctx* ctx, mirt_kernel* k, unsigned dim, const size_t* global, const size_t* local) try {
    (void)local;  // no kernel uses local memory or barriers: the work-group shape is ours to choose
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_enqueue: unknown context");
    if (!live_has(k) || k->ctx != ctx) return fail(ctx, MIRT_E_HANDLE, "mirt_enqueue: unknown kernel");
    const KernelSpec& S = *k->spec;
    if (!global || dim < 1 || dim > 3) return fail(ctx, MIRT_E_ARG, "%s: bad NDRange", S.name);
    for (size_t i = 0; i < k->args.size(); ++i) {
        if (!k->args[i].set) return fail(ctx, MIRT_E_UNSET, "%s: argument %zu was never set (INVALID_KERNEL_ARGS)", S.name, i);
        if (S.args[i] == A_BUF && !live_has(k->args[i].buf)) return fail(ctx, MIRT_E_HANDLE, "%s: argument %zu: buffer was released", S.name, i);
        if (S.args[i] == A_BUF) pin(ctx, k->args[i].buf, false);
    }
    for (unsigned d = 0; d < dim; ++d)
        if (global[d] > 0xFFFFFFFFull) return fail(ctx, MIRT_E_ARG, "%s: global size exceeds 2^32", S.name);
    if (ctx->fusion >= 2 && !ctx->capturing && is_pass_kernel(S.id)) {
        if (S.id == K_initTrace) FLUSH_PENDING(ctx);                 // a new pass begins: whatever was held is not one
        if (S.id == K_initTrace || !ctx->pending.empty()) {
            mirt_ctx::Pending p;
            p.spec = &S; p.args = k->args; p.dim = dim;
            for (unsigned d = 0; d < 3; ++d) p.g[d] = d < dim ? global[d] : 1;
            ctx->pending.push_back(std::move(p));
            if (S.id == K_copyToPixel) return flush_pending(ctx);    // the pass is complete: run it now
            return MIRT_OK;
        }
    }
    FLUSH_PENDING(ctx);
    return launch_kernel(ctx, S, k->args, dim, global);
} MIRT_CATCH("mirt_enqueue", return MIRT_E_DEVICE)

static int launch_kernel(mirt_ctx* ctx, const KernelSpec& S, std::vector<KArg>& a, unsigned dim, const size_t* global) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t g0 = (uint32_t)global[0];
    int rc;
#define BUF(i) (a[i].buf)
#define U(i) (a[i].val.u)
#define F(i) (a[i].val.f)
#define V(i) (a[i].val.v)
    switch (S.id) {
        case K_sizeofRay:
        case K_sizeofPoi:
            if ((rc = need(ctx, S.name, BUF(0), 4))) return rc;
            pt::launch_sizeof(st, S.id == K_sizeofRay, (uint32_t*)BUF(0)->ptr);
            break;
        case K_initAcu: {
            uint32_t cnt = std::min(g0, U(1));
            if ((rc = need(ctx, S.name, BUF(0), (uint64_t)cnt * kAcuBytes))) return rc;
            pt::launch_initAcu(st, BUF(0)->ptr, U(1), g0);
            BUF(0)->version++;
            break;
        }
        case K_initTrace: {
            if (dim != 2) return fail(ctx, MIRT_E_ARG, "initTrace is a 2-D NDRange (A10 code.js:1330)");
            const uint32_t g1 = (uint32_t)global[1];
            const uint32_t cols = f2u_host(V(4)[14]), rows = f2u_host(V(4)[15]);
            const uint32_t rpp = U(7);
            const uint32_t wc = std::min(g0, cols), wr = std::min(g1, rows);
            if (wc && wr) {
                if (rpp == 0) return fail(ctx, MIRT_E_ARG, "initTrace: rays_per_pixel is 0");
                uint64_t rays = ((uint64_t)cols * (wr - 1) + wc) * rpp;  // one past the last ray touched
                if ((rc = need(ctx, "initTrace rays", BUF(1), rays * kRayBytes))) return rc;
                if ((rc = need(ctx, "initTrace pois", BUF(2), rays * kPoiBytes))) return rc;
                if (rpp == 1) {
                    if ((rc = need(ctx, "initTrace seeds", BUF(0), (uint64_t)wc * 4))) return rc;
                    if ((rc = ensure_scratch(ctx, (size_t)cols * rows * 8))) return rc;
                    pt::launch_lensDraws(st, BUF(0)->ptr, ctx->scratch, cols, rows, g0, g1, 0, rows);
                }
                pt::launch_initTrace(st, BUF(1)->ptr, BUF(2)->ptr, ctx->scratch, V(3), V(4), F(5), F(6), rpp, g0, g1);
            }
            break;
        }
        case K_sphereTrace: {
            uint32_t cnt = std::min(g0, U(0));
            if ((rc = need(ctx, "sphereTrace pois", BUF(1), (uint64_t)cnt * kPoiBytes))) return rc;
            if ((rc = need(ctx, "sphereTrace rays", BUF(2), (uint64_t)cnt * kRayBytes))) return rc;
            if ((rc = check_grid(ctx, "sphereTrace grid", BUF(5), U(7), BUF(3), 16, nullptr, BUF(4)))) return rc;
            pt::launch_closest(st, pt::KIND_SPHERES, U(0), BUF(1)->ptr, BUF(2)->ptr, BUF(3)->ptr, nullptr, BUF(4)->ptr, 0, BUF(5)->ptr, V(6), U(7), exit_is_far_face(V(6), U(7)), g0);
            break;
        }
        case K_triangleTrace: {
            uint32_t cnt = std::min(g0, U(0));
            if ((rc = need(ctx, "triangleTrace pois", BUF(1), (uint64_t)cnt * kPoiBytes))) return rc;
            if ((rc = need(ctx, "triangleTrace rays", BUF(2), (uint64_t)cnt * kRayBytes))) return rc;
            if ((rc = check_grid(ctx, "triangleTrace grid", BUF(6), U(8), BUF(3), 48, BUF(4), BUF(5)))) return rc;
            if ((rc = ensure_prepared(ctx, BUF(3), BUF(6)->off_last))) return rc;
            pt::launch_closest(st, pt::KIND_TRIANGLES, U(0), BUF(1)->ptr, BUF(2)->ptr, BUF(3)->prep, BUF(4)->ptr, BUF(5)->ptr, 0, BUF(6)->ptr, V(7), U(8), exit_is_far_face(V(7), U(8)), g0);
            break;
        }
        case K_meshTrace: {
            uint32_t cnt = std::min(g0, U(0));
            if ((rc = need(ctx, "meshTrace pois", BUF(1), (uint64_t)cnt * kPoiBytes))) return rc;
            if ((rc = need(ctx, "meshTrace rays", BUF(2), (uint64_t)cnt * kRayBytes))) return rc;
            if ((rc = check_grid(ctx, "meshTrace grid", BUF(5), U(8), BUF(3), 48, BUF(4), nullptr))) return rc;
            if ((rc = ensure_prepared(ctx, BUF(3), BUF(5)->off_last))) return rc;
            pt::launch_closest(st, pt::KIND_TRIANGLES, U(0), BUF(1)->ptr, BUF(2)->ptr, BUF(3)->prep, BUF(4)->ptr, nullptr, U(6), BUF(5)->ptr, V(7), U(8), exit_is_far_face(V(7), U(8)), g0);
            break;
        }
        case K_lightRender: {
            uint32_t cnt = std::min(g0, U(4));
            if ((rc = need(ctx, "lightRender pois", BUF(0), (uint64_t)cnt * kPoiBytes))) return rc;
            if ((rc = need(ctx, "lightRender rays", BUF(1), (uint64_t)cnt * kRayBytes))) return rc;
            if ((rc = need(ctx, "lightRender acu", BUF(2), (uint64_t)cnt * kAcuBytes))) return rc;
            pt::launch_lightRender(st, BUF(0)->ptr, BUF(1)->ptr, BUF(2)->ptr, V(3), U(4), g0);
            break;
        }
        case K_initShadowTrace: {
            uint32_t cnt = std::min(g0, U(2));
            if ((rc = need(ctx, "initShadowTrace shadow", BUF(0), (uint64_t)cnt * kRayBytes))) return rc;
            if ((rc = need(ctx, "initShadowTrace pois", BUF(1), (uint64_t)cnt * kPoiBytes))) return rc;
            if ((rc = need(ctx, "initShadowTrace seeds", BUF(4), (uint64_t)cnt * 4))) return rc;
            pt::launch_initShadowTrace(st, BUF(0)->ptr, BUF(1)->ptr, U(2), V(3), BUF(4)->ptr, g0);
            break;
        }
        case K_sphereShadowTrace: {
            uint32_t cnt = std::min(g0, U(0));
            if ((rc = need(ctx, "sphereShadowTrace shadow", BUF(1), (uint64_t)cnt * kRayBytes))) return rc;
            if ((rc = check_grid(ctx, "sphereShadowTrace grid", BUF(3), U(5), BUF(2), 16, nullptr, nullptr))) return rc;
            pt::launch_anyhit(st, pt::KIND_SPHERES, U(0), BUF(1)->ptr, BUF(2)->ptr, BUF(3)->ptr, V(4), U(5), exit_is_far_face(V(4), U(5)), g0);
            break;
        }
        case K_triangleShadowTrace: {
            uint32_t cnt = std::min(g0, U(0));
            if ((rc = need(ctx, "triangleShadowTrace shadow", BUF(1), (uint64_t)cnt * kRayBytes))) return rc;
            if ((rc = check_grid(ctx, "triangleShadowTrace grid", BUF(3), U(5), BUF(2), 48, nullptr, nullptr))) return rc;
            if ((rc = ensure_prepared(ctx, BUF(2), BUF(3)->off_last))) return rc;
            pt::launch_anyhit(st, pt::KIND_TRIANGLES, U(0), BUF(1)->ptr, BUF(2)->prep, BUF(3)->ptr, V(4), U(5), exit_is_far_face(V(4), U(5)), g0);
            break;
        }
        case K_sceneRender: {
            uint32_t cnt = std::min(g0, U(5));
            if ((rc = need(ctx, "sceneRender acu", BUF(0), (uint64_t)cnt * kAcuBytes))) return rc;
            if ((rc = need(ctx, "sceneRender pois", BUF(1), (uint64_t)cnt * kPoiBytes))) return rc;
            if ((rc = need(ctx, "sceneRender shadow", BUF(2), (uint64_t)cnt * kRayBytes))) return rc;
            pt::launch_sceneRender(st, BUF(0)->ptr, BUF(1)->ptr, BUF(2)->ptr, BUF(3)->ptr, (uint32_t)(BUF(3)->bytes / 16), V(4), U(5), g0);
            break;
        }
        case K_bouncePaths: {
            uint32_t cnt = std::min(g0, U(3));
            if ((rc = need(ctx, "bouncePaths pois", BUF(0), (uint64_t)cnt * kPoiBytes))) return rc;
            if ((rc = need(ctx, "bouncePaths rays", BUF(1), (uint64_t)cnt * kRayBytes))) return rc;
            if ((rc = need(ctx, "bouncePaths seeds", BUF(2), (uint64_t)cnt * 4))) return rc;
            pt::launch_bouncePaths(st, BUF(0)->ptr, BUF(1)->ptr, BUF(2)->ptr, U(3), g0);
            break;
        }
        case K_copyToPixel: {
            uint32_t cnt = std::min(g0, U(3));
            if ((rc = need(ctx, "copyToPixel pixel", BUF(0), (uint64_t)cnt * 4))) return rc;
            if ((rc = need(ctx, "copyToPixel acu", BUF(1), (uint64_t)cnt * U(4) * kAcuBytes))) return rc;
            pt::launch_copyToPixel(st, BUF(0)->ptr, BUF(1)->ptr, F(2), U(3), U(4), g0, nullptr);
            break;
        }
        case K_a04_sizeofRay:
        case K_a07_sizeofRay:
            if ((rc = need(ctx, S.name, BUF(0), 4))) return rc;
            pt::launch_sizeof(st, true, (uint32_t*)BUF(0)->ptr);
            break;
        case K_a01_raytrace: {
            if (dim != 2) return fail(ctx, MIRT_E_ARG, "raytrace is a 2-D NDRange (A01 code.js:241)");
            const uint32_t g1 = (uint32_t)global[1];
            const uint32_t rows = f2u_host(V(1)[14]), cols = f2u_host(V(1)[15]);   // A01 packs rows, cols (code.js:50)
            const uint32_t wc = std::min(g0, cols), wr = std::min(g1, rows);
            if (wc && wr) {
                if ((rc = need(ctx, "raytrace pixels", BUF(0), ((uint64_t)cols * (wr - 1) + wc) * 4))) return rc;
                pt::launch_a01_raytrace(st, BUF(0)->ptr, V(1), g0, g1);
            }
            break;
        }
        case K_a04_initTrace:
        case K_a07_initTrace:
        case K_a04_meshTrace:
        case K_a07_meshTrace:
        case K_a07_molTrace: {
            if (dim != 2) return fail(ctx, MIRT_E_ARG, "%s is a 2-D NDRange", S.name);
            const uint32_t g1 = (uint32_t)global[1];
            const uint32_t cols = f2u_host(V(1)[14]), rows = f2u_host(V(1)[15]);
            const uint32_t wc = std::min(g0, cols), wr = std::min(g1, rows);
            if (!wc || !wr) break;
            const uint64_t npx = (uint64_t)cols * (wr - 1) + wc;
            if ((rc = need(ctx, "pixels", BUF(0), npx * 4))) return rc;
            if ((rc = need(ctx, "rays", BUF(2), npx * kRayBytes))) return rc;
            if (S.id == K_a04_initTrace) pt::launch_frame_initTrace(st, false, BUF(0)->ptr, V(1), BUF(2)->ptr, nullptr, g0, g1);
            else if (S.id == K_a07_initTrace) pt::launch_frame_initTrace(st, true, BUF(0)->ptr, V(1), BUF(2)->ptr, V(3), g0, g1);
            else if (S.id == K_a04_meshTrace) {
                const uint32_t T = U(3);
                if ((rc = need(ctx, "meshTrace t_pos", BUF(4), (uint64_t)T * 48))) return rc;
                if ((rc = need(ctx, "meshTrace t_normal", BUF(5), (uint64_t)T * 48))) return rc;
                if ((rc = need(ctx, "meshTrace t_mindex", BUF(6), (uint64_t)T * 4))) return rc;
                if ((rc = need(ctx, "meshTrace m_color", BUF(7), 16))) return rc;
                if ((rc = ensure_prepared(ctx, BUF(4), T))) return rc;
                pt::launch_a04_meshTrace(st, BUF(0)->ptr, V(1), BUF(2)->ptr, T, BUF(4)->prep, BUF(5)->ptr, BUF(6)->ptr, BUF(7)->ptr,
                                         (uint32_t)(BUF(7)->bytes / 16), g0, g1);
            } else if (S.id == K_a07_molTrace) {
                // s_mindex / m_color (args 5, 6) are bound by the reference host but never read by the kernel (code.cl:459-460)
                if ((rc = check_grid(ctx, "molTrace grid", BUF(9), U(8), BUF(4), 16, nullptr, nullptr))) return rc;
                pt::launch_a07_molTrace(st, BUF(0)->ptr, V(1), BUF(2)->ptr, BUF(4)->ptr, V(7), U(8), BUF(9)->ptr, g0, g1);
            } else {
                if ((rc = check_grid(ctx, "meshTrace grid", BUF(10), U(9), BUF(4), 48, BUF(5), nullptr))) return rc;
                if ((rc = ensure_prepared(ctx, BUF(4), BUF(10)->off_last))) return rc;
                pt::launch_a07_meshTrace(st, BUF(0)->ptr, V(1), BUF(2)->ptr, BUF(4)->prep, BUF(5)->ptr, V(8), U(9), BUF(10)->ptr, BUF(10)->off_last, g0, g1);
            }
            break;
        }
        default:
            return fail(ctx, MIRT_E_NAME, "kernel not implemented");
    }
#undef BUF
#undef U
#undef F
#undef V
    HIPCHK(ctx, hipGetLastError());
    return MIRT_OK;
}

static int fill_grid(mirt_ctx* ctx, const char* what, const mirt_grid* g, bool tri, bool per_prim_matid, pt::GridArgs* o) {
    if (!g->prims || !g->cell_offsets) return fail(ctx, MIRT_E_ARG, "%s: null geometry buffer", what);
    if (tri && !g->normals) return fail(ctx, MIRT_E_ARG, "%s: triangles need a normal buffer", what);
    if (per_prim_matid && !g->matid) return fail(ctx, MIRT_E_ARG, "%s: needs a per-primitive material buffer", what);
    int rc = check_grid(ctx, what, g->cell_offsets, g->n_slabs, g->prims, tri ? 48 : 16, tri ? g->normals : nullptr,
                        per_prim_matid ? g->matid : nullptr);
    if (rc) return rc;
    o->prims = g->prims->ptr;
    o->kind = tri ? pt::KIND_TRIANGLES : pt::KIND_SPHERES;
    if (tri) {
        mirt_buf* pb = g->prims;
        if ((rc = ensure_prepared(ctx, pb, g->cell_offsets->off_last))) return rc;
        o->prims = pb->prep ? pb->prep : pb->ptr;
        o->pnorm = pb->prep && g->cell_offsets->off_last <= pt::kLdsTriMax ? (const char*)pb->prep + pt::prepared_planes_offset(g->cell_offsets->off_last) : nullptr;
    }
    o->normals = tri ? g->normals->ptr : nullptr;
    o->matid = per_prim_matid ? g->matid->ptr : nullptr;
    o->off = g->cell_offsets->ptr;
    memcpy(o->bound, g->bounds, sizeof o->bound);
    o->n = g->n_slabs;
    o->mesh_matid = g->mesh_matid;
    o->exit_is_far_face = exit_is_far_face(g->bounds, g->n_slabs);
    o->fast_ok = 1;
    for (int k = 0; k < 8; ++k) {
        if ((k & 3) == 3) continue;
        const float a = std::fabs(g->bounds[k]);
        if (!(a == 0.0f || (a >= 9.3132257e-10f && a <= 1048576.0f))) o->fast_ok = 0;
    }
    // the optimistic kernel's box test takes min / max of the two plane quotients as near / far: that needs lo <= hi on every axis
    // (an inverted box is a miss in the reference; here it goes to the exact kernel)
    for (int k = 0; k < 3; ++k)
        if (!(g->bounds[k] <= g->bounds[4 + k])) o->fast_ok = 0;
    if (tri && !g->prims->prep_sane) o->fast_ok = 0;
    // the walk's wave-uniform quotients, once, in the arithmetic the kernel would use: fp32, correctly rounded
    o->walk_ok = 1;
    o->nslots = g->cell_offsets->off_last;   // validated by check_grid above
    o->first_zero = g->cell_offsets->off_first == 0u;
    for (int k = 0; k < 3; ++k) {
        const float span = g->bounds[4 + k] - g->bounds[k];
        const float delta = span / (float)g->n_slabs;
        o->delta[k] = delta;
        o->rdelta[k] = 1.0f / delta;
        const float as = std::fabs(span), ad = std::fabs(delta);
        const bool span_in = span == 0.0f || (as >= 8.6736174e-19f && as <= 1.1529215e18f);   // 0 | 2^-60 .. 2^60 (pt_trace.hpp num_window)
        const bool delta_in = ad >= 9.094947e-13f && ad <= 1.0995116e12f;                      // 2^-40 .. 2^40       (den_window)
        if (!(span_in && delta_in)) o->walk_ok = 0;
    }
    return MIRT_OK;
}

static int render_pass_impl(mirt_ctx* ctx, const mirt_pass_desc* d, bool fresh) {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_render_pass: unknown context");
    if (!d || d->struct_size != sizeof(mirt_pass_desc)) return fail(ctx, MIRT_E_ARG, "mirt_render_pass: descriptor size mismatch");
    if (!d->width || !d->height || !d->rays_per_pixel) return fail(ctx, MIRT_E_ARG, "mirt_render_pass: empty image");
    {   // the host only ever makes k x k rays per pixel (A10 code.js:540); initTrace's k x k loops leave the tail of any other count
        // unwritten (code.cl:479-512), i.e. those rays would be whatever the buffer held -- there is nothing to reproduce
        const uint32_t k = (uint32_t)std::sqrt((double)d->rays_per_pixel);
        const uint32_t kk = (k + 1) * (k + 1) == d->rays_per_pixel ? k + 1 : k;
        if (kk * kk != d->rays_per_pixel)
            return fail(ctx, MIRT_E_ARG, "mirt_render_pass: rays_per_pixel %u is not a square (the lens grid is k x k, A10 code.js:540)", d->rays_per_pixel);
    }
    if (d->n_lights > MIRT_MAX_LIGHTS) return fail(ctx, MIRT_E_ARG, "mirt_render_pass: %u lights > %d (enqueue the kernels one by one instead)", d->n_lights, MIRT_MAX_LIGHTS);
    if (d->n_meshes > MIRT_MAX_MESHES) return fail(ctx, MIRT_E_ARG, "mirt_render_pass: %u meshes > %d (enqueue the kernels one by one instead)", d->n_meshes, MIRT_MAX_MESHES);
    if ((d->n_lights && !d->lights) || (d->n_meshes && !d->meshes)) return fail(ctx, MIRT_E_ARG, "mirt_render_pass: null light/mesh array");
    if (!d->pass_index) return fail(ctx, MIRT_E_ARG, "mirt_render_pass: pass_index is 1-based");
    const uint32_t cols = f2u_host(d->cam[14]), rows = f2u_host(d->cam[15]);
    if (cols != d->width || rows != d->height) return fail(ctx, MIRT_E_ARG, "mirt_render_pass: camera says %ux%u, descriptor %ux%u", cols, rows, d->width, d->height);
    const uint32_t nrows = d->nrows ? d->nrows : d->height;
    if (d->row0 >= d->height || nrows > d->height - d->row0) return fail(ctx, MIRT_E_ARG, "mirt_render_pass: row tile [%u,+%u) outside the image", d->row0, nrows);
    const uint64_t npix = (uint64_t)nrows * d->width;
    const uint64_t nrays = npix * d->rays_per_pixel;
    if (nrays > 0xFFFFFF00ull) return fail(ctx, MIRT_E_ARG, "mirt_render_pass: %llu rays in one tile (the kernels index a tile's rays with 32 bits); split the rows over more launches", (unsigned long long)nrays);
    HIPCHK(ctx, hipSetDevice(ctx->device));

    pt::FusedArgs A;
    memset(&A, 0, sizeof A);
    memcpy(A.cam, d->cam, sizeof A.cam);
    memcpy(A.bound, d->scene_bounds, sizeof A.bound);
    A.focal_length = d->focal_length; A.lens_rad = d->lens_rad;
    A.width = d->width; A.height = d->height; A.rpp = d->rays_per_pixel;
    A.row0 = d->row0; A.nrows = nrows; A.bounces = d->bounces;
    A.n_lights = d->n_lights;
    A.fresh = fresh ? 1u : 0u;
    int rc;
    if (d->spheres && (rc = fill_grid(ctx, "spheres", d->spheres, false, true, &A.sets[A.n_sets++]))) return rc;
    if (d->triangles && (rc = fill_grid(ctx, "triangles", d->triangles, true, true, &A.sets[A.n_sets++]))) return rc;
    for (uint32_t m = 0; m < d->n_meshes; ++m)
        if ((rc = fill_grid(ctx, "mesh", &d->meshes[m], true, false, &A.sets[A.n_sets++]))) return rc;
    // one prepared copy per position buffer, laid out for ONE record count: two sets of a pass that share a buffer with different counts would have
    // the second fill_grid re-lay the copy the first set's pointers describe
    {
        const mirt_grid* gs[2 + MIRT_MAX_MESHES];
        uint32_t ng = 0;
        if (d->triangles) gs[ng++] = d->triangles;
        for (uint32_t m = 0; m < d->n_meshes; ++m) gs[ng++] = &d->meshes[m];
        for (uint32_t i = 0; i < ng; ++i)
            for (uint32_t j = i + 1; j < ng; ++j)
                if (gs[i]->prims == gs[j]->prims && gs[i]->cell_offsets->off_last != gs[j]->cell_offsets->off_last)
                    return fail(ctx, MIRT_E_ARG, "mirt_render_pass: two triangle sets share a position buffer but hold %u and %u slots (one prepared copy per buffer: "
                                                 "give each set its own buffer)", gs[i]->cell_offsets->off_last, gs[j]->cell_offsets->off_last);
    }
    for (uint32_t l = 0; l < d->n_lights; ++l) {
        memcpy(A.lights[l].shadow, d->lights[l].shadow, 64);
        memcpy(A.lights[l].scene, d->lights[l].scene, 64);
        memcpy(A.lights[l].light, d->lights[l].light, 64);
    }
    if ((rc = need(ctx, "material", d->material, 16))) return rc;
    A.material = d->material->ptr;
    A.nmat = (uint32_t)(d->material->bytes / 16);
    if ((rc = need(ctx, "seeds", d->seeds, nrays * 4))) return rc;
    // copyToPixel inside the pass: a frame's first pass at a ray count that puts whole pixels into a block of 256 ray ids (pt_launch.hpp
    // fused_resolves).  Then -- and only then -- `acu` is optional: without it nothing per ray but the seed touches memory.
    const bool resolve_in_pass = ctx->inpass_resolve && pt::fused_resolves(d->rays_per_pixel, d->pixel || d->radiance) && (fresh || d->acu);
    ctx->last_pass_resolved = resolve_in_pass;
    if (!d->acu && !(resolve_in_pass && fresh))
        return fail(ctx, MIRT_E_ARG, "mirt_render_pass: acu may only be null for a frame's first pass (mirt_render_first_pass) with a pixel or radiance buffer and "
                                     "rays_per_pixel dividing 256 or 256 times a power of two up to 32 (here: %s, %u rays per pixel%s)", fresh ? "first pass" : "NOT a first pass",
                    d->rays_per_pixel, d->pixel || d->radiance ? "" : ", no output buffer");
    if (d->acu && (rc = need(ctx, "acu", d->acu, nrays * kAcuBytes))) return rc;
    A.seeds = (int32_t*)d->seeds->ptr;
    A.acu = d->acu ? d->acu->ptr : nullptr;
    if (d->pixel && (rc = need(ctx, "pixel", d->pixel, npix * 4))) return rc;
    if (d->radiance && (rc = need(ctx, "radiance", d->radiance, npix * 16))) return rc;
    A.chunks = 1u;
    A.chunk_bits = ~0u;
    if (resolve_in_pass) {
        A.resolve = 1u;
        A.pixel = d->pixel ? d->pixel->ptr : nullptr;
        A.radiance = d->radiance ? d->radiance->ptr : nullptr;
        A.res_m = (float)(1.0 / ((double)d->rays_per_pixel * (double)d->pass_index));  // A10 code.js:1412
        if (ctx->res_m_override == ctx->res_m_override) A.res_m = ctx->res_m_override;   // (not a NaN: try_fuse_pass hands over the recorded copyToPixel's own factor)
        A.chunks = pt::fused_chunks(d->rays_per_pixel);
        if (A.chunks > 1u && !A.radiance) {   // a pixel of more than 256 rays: its sums travel from launch to launch through memory (FusedArgs::chunks)
            if ((rc = ensure_scratch(ctx, (size_t)npix * 16))) return rc;
            A.radiance = ctx->scratch;
        }
    }

    if (A.rpp == 1) {
        // the column streams live in seeds[0..width): only the tile that owns row 0 holds them
        if (d->row0 != 0) return fail(ctx, MIRT_E_ARG, "mirt_render_pass: rays_per_pixel == 1 couples rows through seeds[col] (A10 code.cl:429); render it as one tile");
        if ((rc = ensure_scratch(ctx, (size_t)npix * 8))) return rc;
        pt::launch_lensDraws(ctx->stream, A.seeds, ctx->scratch, d->width, d->height, d->width, d->height, d->row0, nrows);
        A.uv = ctx->scratch;
    }
    if (ctx->profiling && !ctx->capturing) HIPCHK(ctx, hipEventRecord(ctx->pe[0], ctx->stream));
    bool optimistic = pt::fused_fast_available() && !ctx->force_exact;
    for (uint32_t i = 0; i < A.n_sets; ++i) optimistic = optimistic && A.sets[i].fast_ok != 0;
    if (optimistic) {
        // optimistic kernel (exact cheap divisions inside their window) + exact kernel over the samples that left the window:
        // the second launch walks the first one's bit mask on the device, so the pair is queued without a host round trip
        // one bit per sample -- or, resolving in the pass, per block of 256 ray ids (pt_kernels_fused.hip)
        const uint32_t words = resolve_in_pass ? (uint32_t)(((nrays + 255) / 256 + 31) / 32) : (uint32_t)((nrays + 31) / 32);
        const size_t need_bytes = 16 + (size_t)words * 4;
        if (ctx->defer_bytes < need_bytes) {
            NOT_WHILE_CAPTURING(ctx, "growing the deferred-sample mask");
            if (ctx->defer) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->defer)); ctx->defer = nullptr; ctx->defer_bytes = 0; }
            ctx->defer_gen++;
            HIPCHK(ctx, hipMalloc(&ctx->defer, need_bytes));
            ctx->defer_bytes = need_bytes;
        }
        if (ctx->capturing) ctx->cap_defer = true;
        uint32_t* mask = (uint32_t*)ctx->defer + 4;
        HIPCHK(ctx, hipMemsetAsync(ctx->defer, 0, need_bytes, ctx->stream));
        // one pair of launches -- or, a pixel of more than 256 rays resolved in the pass, one pair per block of 256 rays of a pixel, in ray order: launch c
        // goes on from the sums launch c - 1 and its redo left (FusedArgs::chunks).  The mask collects the deferred blocks of every launch.
        void* const pixel_out = A.pixel;
        for (uint32_t c = 0; c < A.chunks; ++c) {
            A.chunk = c;
            if (A.chunks > 1u) {   // every chunks-th bit from bit c on (chunks divides 32: a block's bit sits at its number modulo 32)
                A.chunk_bits = 0u;
                for (uint32_t b = c; b < 32u; b += A.chunks) A.chunk_bits |= 1u << b;
            }
            A.pixel = c + 1u == A.chunks ? pixel_out : nullptr;
            pt::launch_fused(ctx->stream, A, true, mask, nullptr, 0);
            pt::launch_fused(ctx->stream, A, false, nullptr, mask, words);
        }
        ctx->defer_words = words;
        ctx->defer_unit = resolve_in_pass ? 256u : 1u;
    } else {
        void* const pixel_out = A.pixel;
        for (uint32_t c = 0; c < A.chunks; ++c) {
            A.chunk = c;
            A.pixel = c + 1u == A.chunks ? pixel_out : nullptr;
            pt::launch_fused(ctx->stream, A, false, nullptr, nullptr, 0);
        }
        ctx->defer_words = 0;
    }
    if (ctx->profiling && !ctx->capturing) HIPCHK(ctx, hipEventRecord(ctx->pe[1], ctx->stream));
    if (!resolve_in_pass && (d->pixel || d->radiance)) {
        float m = (float)(1.0 / ((double)d->rays_per_pixel * (double)d->pass_index));  // A10 code.js:1412
        if (ctx->res_m_override == ctx->res_m_override) m = ctx->res_m_override;
        pt::launch_copyToPixel(ctx->stream, d->pixel ? d->pixel->ptr : nullptr, A.acu, m, (uint32_t)npix, A.rpp, (uint32_t)npix,
                               d->radiance ? d->radiance->ptr : nullptr);
    }
    if (ctx->profiling && !ctx->capturing) { HIPCHK(ctx, hipEventRecord(ctx->pe[2], ctx->stream)); ctx->pe_valid = true; }
    HIPCHK(ctx, hipGetLastError());
    d->seeds->version++;
    if (d->acu) d->acu->version++;
    return MIRT_OK;
}

int mirt_render_pass(mirt_ctx* ctx, const mirt_pass_desc* d) try { if (live_has(ctx)) FLUSH_PENDING(ctx); return render_pass_impl(ctx, d, false); } MIRT_CATCH("mirt_render_pass", return MIRT_E_DEVICE)
int mirt_render_first_pass(mirt_ctx* ctx, const mirt_pass_desc* d) try { if (live_has(ctx)) FLUSH_PENDING(ctx); return render_pass_impl(ctx, d, true); } MIRT_CATCH("mirt_render_first_pass", return MIRT_E_DEVICE)

int mirt_ctx_set_fusion(mirt_ctx* ctx, int level) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_ctx_set_fusion: unknown context");
    if (level != 0 && level != 2) return fail(ctx, MIRT_E_ARG, "mirt_ctx_set_fusion: level is 0 (every enqueue launches) or 2 (whole passes are fused)");
    FLUSH_PENDING(ctx);
    ctx->fusion = level;
    return MIRT_OK;
} MIRT_CATCH("mirt_ctx_set_fusion", return MIRT_E_DEVICE)

int mirt_ctx_fused_passes(mirt_ctx* ctx, uint64_t* count) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_ctx_fused_passes: unknown context");
    if (!count) return fail(ctx, MIRT_E_ARG, "mirt_ctx_fused_passes: null output");
    *count = ctx->fused_passes;
    return MIRT_OK;
} MIRT_CATCH("mirt_ctx_fused_passes", return MIRT_E_DEVICE)

int mirt_ctx_set_exact_only(mirt_ctx* ctx, int on) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_ctx_set_exact_only: unknown context");
    FLUSH_PENDING(ctx);
    ctx->force_exact = on != 0;
    return MIRT_OK;
} MIRT_CATCH("mirt_ctx_set_exact_only", return MIRT_E_DEVICE)

int mirt_pass_deferred(mirt_ctx* ctx, uint64_t* samples) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_pass_deferred: unknown context");
    FLUSH_PENDING(ctx);
    NOT_WHILE_CAPTURING(ctx, "mirt_pass_deferred");
    if (!samples) return fail(ctx, MIRT_E_ARG, "mirt_pass_deferred: null output");
    *samples = 0;
    if (!ctx->defer_words) return MIRT_OK;
    // counted on demand: the mask of the last pass is still in place (the next pass clears it)
    uint32_t* counters = (uint32_t*)ctx->defer;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(counters, 0, 16, ctx->stream));
    pt::launch_deferCount(ctx->stream, counters + 4, ctx->defer_words, counters);
    uint32_t count = 0;
    HIPCHK(ctx, hipMemcpyAsync(&count, counters, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *samples = (uint64_t)count * ctx->defer_unit;   // resolving in the pass the exact kernel re-runs whole blocks of 256 samples
    return MIRT_OK;
} MIRT_CATCH("mirt_pass_deferred", return MIRT_E_DEVICE)

int mirt_ctx_set_profiling(mirt_ctx* ctx, int on) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_ctx_set_profiling: unknown context");
    FLUSH_PENDING(ctx);
    ctx->profiling = on != 0;
    ctx->pe_valid = false;
    return MIRT_OK;
} MIRT_CATCH("mirt_ctx_set_profiling", return MIRT_E_DEVICE)

int mirt_pass_timing(mirt_ctx* ctx, float* fused_ms, float* resolve_ms) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_pass_timing: unknown context");
    FLUSH_PENDING(ctx);
    if (!ctx->pe_valid) return fail(ctx, MIRT_E_ARG, "mirt_pass_timing: no profiled mirt_render_pass yet (mirt_ctx_set_profiling)");
    HIPCHK(ctx, hipEventSynchronize(ctx->pe[2]));
    float a = 0.f, b = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&a, ctx->pe[0], ctx->pe[1]));
    HIPCHK(ctx, hipEventElapsedTime(&b, ctx->pe[1], ctx->pe[2]));
    if (fused_ms) *fused_ms = a;
    if (resolve_ms) *resolve_ms = b;
    return MIRT_OK;
} MIRT_CATCH("mirt_pass_timing", return MIRT_E_DEVICE)

int mirt_seed_fill(mirt_ctx* ctx, mirt_buf* seeds, uint64_t first_ray, uint64_t count, uint32_t seed_base) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_seed_fill: unknown context");
    FLUSH_PENDING(ctx);
    int rc = need(ctx, "mirt_seed_fill", seeds, count * 4);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pt::launch_seedFill(ctx->stream, seeds->ptr, first_ray, count, seed_base);
    HIPCHK(ctx, hipGetLastError());
    seeds->version++;
    return MIRT_OK;
} MIRT_CATCH("mirt_seed_fill", return MIRT_E_DEVICE)

int mirt_zero(mirt_ctx* ctx, mirt_buf* buf) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_zero: unknown context");
    FLUSH_PENDING(ctx);
    int rc = need(ctx, "mirt_zero", buf, 0);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(buf->ptr, 0, buf->bytes, ctx->stream));
    buf->version++;
    return MIRT_OK;
} MIRT_CATCH("mirt_zero", return MIRT_E_DEVICE)

int mirt_debug_prepared(mirt_ctx* ctx, mirt_buf* positions, uint32_t count, void* out, size_t bytes, size_t* total) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_debug_prepared: unknown context");
    FLUSH_PENDING(ctx);
    NOT_WHILE_CAPTURING(ctx, "mirt_debug_prepared");
    if (!live_has(positions) || positions->ctx != ctx) return fail(ctx, MIRT_E_HANDLE, "mirt_debug_prepared: unknown buffer");
    int rc = need(ctx, "mirt_debug_prepared positions", positions, (uint64_t)count * 48);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = ensure_prepared(ctx, positions, count))) return rc;
    const size_t have = pt::prepared_bytes(count);
    if (total) *total = have;
    if (out && bytes && positions->prep) {
        HIPCHK(ctx, hipMemcpyAsync(out, positions->prep, bytes < have ? bytes : have, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MIRT_OK;
} MIRT_CATCH("mirt_debug_prepared", return MIRT_E_DEVICE)

int mirt_debug_numerics(mirt_ctx* ctx, int op, mirt_buf* a, mirt_buf* b, mirt_buf* out, size_t n) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_debug_numerics: unknown context");
    FLUSH_PENDING(ctx);
    int rc;
    if (!((op >= 0 && op <= 14) || (op >= 20 && op <= 27))) return fail(ctx, MIRT_E_ARG, "mirt_debug_numerics: op %d outside 0..14, 20..27", op);
    const bool vec = op >= 20 && op != 25;
    const uint64_t in_w = vec ? 3 : 1, out_w = (op == 21 || op == 24) ? 3 : op == 27 ? 4 : 1;
    const bool two = op == 0 || (op >= 6 && op <= 12) || op == 20 || op == 21 || op == 23 || op == 25;
    if ((rc = need(ctx, "mirt_debug_numerics a", a, (uint64_t)n * 4 * in_w))) return rc;
    if (two && !b) return fail(ctx, MIRT_E_ARG, "mirt_debug_numerics: op %d needs two inputs", op);
    if (b && (rc = need(ctx, "mirt_debug_numerics b", b, (uint64_t)n * 4 * (two ? in_w : 1)))) return rc;
    if ((rc = need(ctx, "mirt_debug_numerics out", out, (uint64_t)n * 4 * out_w))) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pt::launch_numerics(ctx->stream, op, a->ptr, b ? b->ptr : nullptr, out->ptr, n);
    HIPCHK(ctx, hipGetLastError());
    return MIRT_OK;
} MIRT_CATCH("mirt_debug_numerics", return MIRT_E_DEVICE)

static int new_owned(mirt_ctx* ctx, size_t bytes, mirt_buf** out) {
    return mirt_buf_create(ctx, bytes ? bytes : 16, MIRT_MEM_READ_WRITE, out);
}

int mirt_grid_build(mirt_ctx* ctx, const mirt_grid_build_desc* d, mirt_buf** cell_offsets, mirt_buf** order, uint32_t* total) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_grid_build: unknown context");
    FLUSH_PENDING(ctx);
    if (!d || d->struct_size != sizeof(mirt_grid_build_desc) || !cell_offsets || !order || !total)
        return fail(ctx, MIRT_E_ARG, "mirt_grid_build: null argument or descriptor size mismatch");
    *cell_offsets = *order = nullptr;
    *total = 0;
    if (d->kind > 1) return fail(ctx, MIRT_E_ARG, "mirt_grid_build: kind is 0 (spheres) or 1 (triangles)");
    if (d->n_slabs == 0 || d->n_slabs > 1024) return fail(ctx, MIRT_E_ARG, "mirt_grid_build: n_slabs %u outside 1..1024", d->n_slabs);
    int rc;
    if (d->count && (rc = need(ctx, "mirt_grid_build prims", d->prims_f64, (uint64_t)d->count * (d->kind ? 72 : 32)))) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t cells = (uint64_t)d->n_slabs * d->n_slabs * d->n_slabs;
    if ((rc = new_owned(ctx, (cells + 1) * 4, cell_offsets))) return rc;
    uint32_t* ord = nullptr;
    uint64_t slots = 0;
    hipError_t e = pt::grid_build(ctx->stream, (int)d->kind, d->count ? (const double*)d->prims_f64->ptr : nullptr, d->count, d->bounds, d->n_slabs,
                                  (uint32_t*)(*cell_offsets)->ptr, &ord, total, &slots);
    if (e != hipSuccess || slots > pt::kMaxGridSlots) {
        mirt_buf_release(*cell_offsets);
        *cell_offsets = nullptr;
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(ctx, MIRT_E_DEVICE, "mirt_grid_build: %s", hipGetErrorString(e)); }
        return fail(ctx, MIRT_E_RANGE, "mirt_grid_build: the grid would hold %llu (cell, primitive) slots; at most %llu are supported -- use fewer slabs",
                    (unsigned long long)slots, (unsigned long long)pt::kMaxGridSlots);
    }
    // adopt the order array as an owned buffer
    mirt_buf* ob = new mirt_buf();
    ob->ctx = ctx; ob->bytes = *total ? (size_t)*total * 4 : 16; ob->owned = true; ob->flags = MIRT_MEM_READ_WRITE; ob->uid = g_next_uid++;
    if (ord) ob->ptr = ord;
    else if (hipMalloc(&ob->ptr, 16) != hipSuccess) { delete ob; return fail(ctx, MIRT_E_DEVICE, "mirt_grid_build: hipMalloc failed"); }
    live_add(ob, H_BUF);
    ctx->bufs.insert(ob);
    *order = ob;
    (*cell_offsets)->version++;
    return MIRT_OK;
} MIRT_CATCH("mirt_grid_build", return MIRT_E_DEVICE)

int mirt_grid_gather_triangles(mirt_ctx* ctx, mirt_buf* order, uint32_t total, mirt_buf* pos_f64, mirt_buf* nor_f64, uint32_t nsteps,
                               const int32_t* ops, const double* vecs, float pad_w, mirt_buf** pos_out, mirt_buf** nor_out) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_grid_gather_triangles: unknown context");
    FLUSH_PENDING(ctx);
    if (!pos_out || nsteps > 4 || (nsteps && (!ops || !vecs))) return fail(ctx, MIRT_E_ARG, "mirt_grid_gather_triangles: bad argument");
    for (uint32_t i = 0; i < nsteps; ++i) if (ops[i] < 0 || ops[i] > 2) return fail(ctx, MIRT_E_ARG, "mirt_grid_gather_triangles: op %d", ops[i]);
    int rc;
    if ((rc = need(ctx, "gather order", order, (uint64_t)total * 4))) return rc;
    if ((rc = need(ctx, "gather positions", pos_f64, 0))) return rc;
    if (nor_f64 && nor_out && (rc = need(ctx, "gather normals", nor_f64, 0))) return rc;
    if (pos_f64->bytes % 72) return fail(ctx, MIRT_E_ARG, "mirt_grid_gather_triangles: positions are 9 doubles per triangle");
    // every index in `order` must address a triangle of the inputs: order comes from mirt_grid_build over `count` primitives;
    // a foreign order array is checked on the host
    if (total) {
        std::vector<uint32_t> h(total);
        HIPCHK(ctx, hipMemcpyAsync(h.data(), order->ptr, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        const uint64_t ntri = pos_f64->bytes / 72;
        for (uint32_t v : h) if (v >= ntri) return fail(ctx, MIRT_E_DATA, "mirt_grid_gather_triangles: order refers to triangle %u of %llu", v, (unsigned long long)ntri);
        if (nor_f64 && nor_out && nor_f64->bytes < ntri * 72) return fail(ctx, MIRT_E_RANGE, "mirt_grid_gather_triangles: normals shorter than positions");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = new_owned(ctx, (size_t)total * 48, pos_out))) return rc;
    const bool want_n = nor_f64 && nor_out;
    if (nor_out) *nor_out = nullptr;
    if (want_n && (rc = new_owned(ctx, (size_t)total * 48, nor_out))) return rc;
    pt::launch_gatherTriangles(ctx->stream, (const uint32_t*)order->ptr, total, (const double*)pos_f64->ptr, want_n ? (const double*)nor_f64->ptr : nullptr,
                               (int)nsteps, (const int*)ops, vecs, pad_w, (*pos_out)->ptr, want_n ? (*nor_out)->ptr : nullptr);
    HIPCHK(ctx, hipGetLastError());
    (*pos_out)->version++;
    return MIRT_OK;
} MIRT_CATCH("mirt_grid_gather_triangles", return MIRT_E_DEVICE)

int mirt_mesh_ingest(mirt_ctx* ctx, const mirt_mesh_ingest_desc* d, mirt_buf* pos9_out, mirt_buf* nor9_out, mirt_buf* bounds6) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_mesh_ingest: unknown context");
    FLUSH_PENDING(ctx);
    if (!d || d->struct_size != sizeof(mirt_mesh_ingest_desc)) return fail(ctx, MIRT_E_ARG, "mirt_mesh_ingest: null descriptor or size mismatch");
    NOT_WHILE_CAPTURING(ctx, "mirt_mesh_ingest");
    if (d->n_corners % 3u) return fail(ctx, MIRT_E_ARG, "mirt_mesh_ingest: %u corners is not a whole number of triangles", d->n_corners);
    int rc;
    if ((rc = need(ctx, "mirt_mesh_ingest positions", d->positions_f64, (uint64_t)d->n_vertices * 24))) return rc;
    if ((rc = need(ctx, "mirt_mesh_ingest normals", d->normals_f64, (uint64_t)d->n_vertices * 24))) return rc;
    if (d->indices_u32 && (rc = need(ctx, "mirt_mesh_ingest indices", d->indices_u32, (uint64_t)d->n_corners * 4))) return rc;
    if (!d->indices_u32 && d->n_corners > d->n_vertices) return fail(ctx, MIRT_E_RANGE, "mirt_mesh_ingest: %u corners of an un-indexed mesh with %u vertices", d->n_corners, d->n_vertices);
    const uint64_t end = ((uint64_t)d->first_corner + d->n_corners) * 24;
    if ((rc = need(ctx, "mirt_mesh_ingest positions out", pos9_out, end))) return rc;
    if ((rc = need(ctx, "mirt_mesh_ingest normals out", nor9_out, end))) return rc;
    if ((rc = need(ctx, "mirt_mesh_ingest bounds", bounds6, 24))) return rc;
    if ((rc = ensure_scratch(ctx, 32))) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(ctx->scratch, 0, 32, ctx->stream));
    pt::launch_meshIngest(ctx->stream, (const double*)d->positions_f64->ptr, (const double*)d->normals_f64->ptr,
                          d->indices_u32 ? (const uint32_t*)d->indices_u32->ptr : nullptr, d->n_vertices, d->n_corners, d->model, d->normal_mat,
                          (double*)pos9_out->ptr + 3u * (size_t)d->first_corner, (double*)nor9_out->ptr + 3u * (size_t)d->first_corner,
                          (float*)bounds6->ptr, (uint32_t*)ctx->scratch);
    uint32_t flag = 0;
    HIPCHK(ctx, hipMemcpyAsync(&flag, (const char*)ctx->scratch + 24, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    pos9_out->version++; nor9_out->version++; bounds6->version++;
    if (flag) return fail(ctx, MIRT_E_DATA, "mirt_mesh_ingest: an index refers past the %u vertices of the mesh", d->n_vertices);
    return MIRT_OK;
} MIRT_CATCH("mirt_mesh_ingest", return MIRT_E_DEVICE)

static int check_order(mirt_ctx* ctx, mirt_buf* order, uint32_t total, uint64_t n_in) {
    if (!total) return MIRT_OK;
    std::vector<uint32_t> h(total);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), order->ptr, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t v : h) if (v >= n_in) return fail(ctx, MIRT_E_DATA, "order refers to element %u of %llu", v, (unsigned long long)n_in);
    return MIRT_OK;
}

int mirt_grid_gather_spheres(mirt_ctx* ctx, mirt_buf* order, uint32_t total, mirt_buf* sph_f64, mirt_buf** out) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_grid_gather_spheres: unknown context");
    FLUSH_PENDING(ctx);
    if (!out) return fail(ctx, MIRT_E_ARG, "mirt_grid_gather_spheres: null out");
    int rc;
    if ((rc = need(ctx, "gather order", order, (uint64_t)total * 4))) return rc;
    if ((rc = need(ctx, "gather spheres", sph_f64, 0))) return rc;
    if ((rc = check_order(ctx, order, total, sph_f64->bytes / 32))) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = new_owned(ctx, (size_t)total * 16, out))) return rc;
    pt::launch_gatherSpheres(ctx->stream, (const uint32_t*)order->ptr, total, (const double*)sph_f64->ptr, (*out)->ptr);
    HIPCHK(ctx, hipGetLastError());
    (*out)->version++;
    return MIRT_OK;
} MIRT_CATCH("mirt_grid_gather_spheres", return MIRT_E_DEVICE)

int mirt_grid_gather_u32(mirt_ctx* ctx, mirt_buf* order, uint32_t total, mirt_buf* in_u32, mirt_buf** out) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_grid_gather_u32: unknown context");
    FLUSH_PENDING(ctx);
    if (!out) return fail(ctx, MIRT_E_ARG, "mirt_grid_gather_u32: null out");
    int rc;
    if ((rc = need(ctx, "gather order", order, (uint64_t)total * 4))) return rc;
    if ((rc = need(ctx, "gather input", in_u32, 0))) return rc;
    if ((rc = check_order(ctx, order, total, in_u32->bytes / 4))) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = new_owned(ctx, (size_t)total * 4, out))) return rc;
    pt::launch_gatherU32(ctx->stream, (const uint32_t*)order->ptr, total, (const uint32_t*)in_u32->ptr, (uint32_t*)(*out)->ptr);
    HIPCHK(ctx, hipGetLastError());
    (*out)->version++;
    return MIRT_OK;
} MIRT_CATCH("mirt_grid_gather_u32", return MIRT_E_DEVICE)

int mirt_debug_divcheck(mirt_ctx* ctx, int mode, uint64_t seed, uint64_t count, mirt_buf* out16) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_debug_divcheck: unknown context");
    FLUSH_PENDING(ctx);
    int rc = need(ctx, "mirt_debug_divcheck out", out16, 16 * 8);
    if (rc) return rc;
    if (mode < 0 || mode > 5) return fail(ctx, MIRT_E_ARG, "mirt_debug_divcheck: mode is 0..5");
    if (mode == 4 && (seed > (1u << 23) || count > (1u << 23) - seed)) return fail(ctx, MIRT_E_ARG, "mirt_debug_divcheck: mode 4 walks denominators [seed, seed+count) within 2^23 mantissas");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(out16->ptr, 0, 16 * 8, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync((char*)out16->ptr + 80, 0xFF, 8, ctx->stream));   // out[10]: running minimum
    pt::launch_divCheck(ctx->stream, mode, seed, count, out16->ptr);
    HIPCHK(ctx, hipGetLastError());
    return MIRT_OK;
} MIRT_CATCH("mirt_debug_divcheck", return MIRT_E_DEVICE)

int mirt_capture_begin(mirt_ctx* ctx) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_capture_begin: unknown context");
    FLUSH_PENDING(ctx);
    if (ctx->capturing) return fail(ctx, MIRT_E_ARG, "mirt_capture_begin: already recording");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    ctx->capturing = true;
    ctx->cap_pins.clear();
    ctx->cap_scratch = ctx->cap_defer = false;
    return MIRT_OK;
} MIRT_CATCH("mirt_capture_begin", return MIRT_E_DEVICE)

int mirt_capture_end(mirt_ctx* ctx, mirt_graph** out) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_capture_end: unknown context");
    if (!out) return fail(ctx, MIRT_E_ARG, "mirt_capture_end: null output");
    if (!ctx->capturing) return fail(ctx, MIRT_E_ARG, "mirt_capture_end: not recording");
    ctx->capturing = false;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (e != hipSuccess || !g) {
        (void)hipGetLastError();
        return fail(ctx, MIRT_E_DEVICE, "mirt_capture_end: the recording was invalidated (%s)", hipGetErrorString(e));
    }
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        return fail(ctx, MIRT_E_DEVICE, "mirt_capture_end: hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    mirt_graph* mg = new mirt_graph;
    mg->ctx = ctx; mg->graph = g; mg->exec = x;
    mg->pins.swap(ctx->cap_pins);
    mg->scratch_gen = ctx->cap_scratch ? ctx->scratch_gen : 0;
    mg->defer_gen = ctx->cap_defer ? ctx->defer_gen : 0;
    live_add(mg, H_GRAPH);
    ctx->graphs.insert(mg);
    *out = mg;
    return MIRT_OK;
} MIRT_CATCH("mirt_capture_end", return MIRT_E_DEVICE)

int mirt_graph_launch(mirt_ctx* ctx, mirt_graph* graph) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_graph_launch: unknown context");
    FLUSH_PENDING(ctx);
    if (!live_has(graph) || graph->ctx != ctx) return fail(ctx, MIRT_E_HANDLE, "mirt_graph_launch: unknown graph");
    NOT_WHILE_CAPTURING(ctx, "mirt_graph_launch");
    // the recording holds raw device pointers: refuse to replay once any of them has been freed, replaced or (for validated /
    // prepared geometry) rewritten since -- record the sequence again
    for (const auto& p : graph->pins) {
        if (!live_has(p.buf) || p.buf->uid != p.uid)
            return fail(ctx, MIRT_E_HANDLE, "mirt_graph_launch: a buffer the recording uses has been released; record the sequence again");
        if (p.content && (p.buf->version != p.version || p.buf->prep_gen != p.prep_gen))
            return fail(ctx, MIRT_E_HANDLE, "mirt_graph_launch: geometry the recording was validated against has been rewritten; record the sequence again");
    }
    if ((graph->scratch_gen && graph->scratch_gen != ctx->scratch_gen) || (graph->defer_gen && graph->defer_gen != ctx->defer_gen))
        return fail(ctx, MIRT_E_HANDLE, "mirt_graph_launch: the context's scratch memory was reallocated after the recording; record the sequence again");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipGraphLaunch(graph->exec, ctx->stream));
    return MIRT_OK;
} MIRT_CATCH("mirt_graph_launch", return MIRT_E_DEVICE)

int mirt_graph_release(mirt_graph* graph) try {
    if (!live_has(graph)) return fail(nullptr, MIRT_E_HANDLE, "mirt_graph_release: unknown graph");
    free_graph(graph, live_has(graph->ctx));
    return MIRT_OK;
} MIRT_CATCH("mirt_graph_release", return MIRT_E_DEVICE)

int mirt_timer_start(mirt_ctx* ctx) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_timer_start: unknown context");
    FLUSH_PENDING(ctx);
    NOT_WHILE_CAPTURING(ctx, "mirt_timer_start");
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return MIRT_OK;
} MIRT_CATCH("mirt_timer_start", return MIRT_E_DEVICE)

int mirt_timer_stop_ms(mirt_ctx* ctx, float* ms) try {
    if (!live_has(ctx)) return fail(nullptr, MIRT_E_HANDLE, "mirt_timer_stop_ms: unknown context");
    FLUSH_PENDING(ctx);
    NOT_WHILE_CAPTURING(ctx, "mirt_timer_stop_ms");
    if (!ms) return fail(ctx, MIRT_E_ARG, "mirt_timer_stop_ms: null output");
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return MIRT_OK;
} MIRT_CATCH("mirt_timer_stop_ms", return MIRT_E_DEVICE)

}  // extern "C"

#if PT_COUNT
// profiling builds only (-DPT_COUNT=1, profiles/trip_counts.py): the fused pass's wave-level trip counters (pt_trace.hpp PtCounter); not in include/mirt.h
namespace pt { int debug_counters(unsigned long long* out, int reset); }
extern "C" MIRT_API int mirt_debug_counters(unsigned long long* out32, int reset) { return pt::debug_counters(out32, reset); }
#endif
