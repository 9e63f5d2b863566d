/*
 * mirt.h -- C ABI of libmirt.so: the MI355X-native device runtime behind the host
 * interface of eaymerich/2015-RayTracing's Assign10 path tracer.
 *
 * The reference host (Assign10-Path_Tracing/code.js, "A10 code.js" below) talks to its
 * device through the WebCL 1.0 object model (webcl -> context -> queue / program ->
 * kernel / buffer).  Each entry point here is what that binding needs underneath; the
 * comment on each cites the reference call it replaces.  Signatures are plain C
 * (opaque handles, pointers, sizes); no C++ exception crosses this boundary.
 *
 * Conventions
 *   - return 0 (MIRT_OK) or a negative mirt_status; text via mirt_last_error().
 *   - host pointers are borrowed for the duration of the call only: reads/writes of
 *     pageable memory complete before the call returns, whatever `blocking` says (the
 *     reference passes blocking=false and keeps its typed arrays alive until finish()).
 *   - one in-order HIP stream per context (A10 code.js:592 createCommandQueue()).
 *   - handles are reference-free: release exactly once; using a released handle -- or a
 *     handle of the wrong kind -- is MIRT_E_HANDLE, not undefined behaviour (every handle is
 *     validated against a live table keyed by address AND kind).
 *   - a context owns what was created on it: mirt_ctx_destroy releases the buffers, kernels
 *     and graphs the host left behind (the reference host never releases its bouncePaths
 *     kernel, A10 code.js:1444-1455); those handles are MIRT_E_HANDLE afterwards.
 *   - single host thread per context (the reference is a single JS thread).
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MIRT_API __attribute__((visibility("default")))
#else
#define MIRT_API
#endif

typedef struct mirt_ctx mirt_ctx;
typedef struct mirt_buf mirt_buf;
typedef struct mirt_kernel mirt_kernel;

typedef enum mirt_status {
    MIRT_OK = 0,
    MIRT_E_ARG = -1,        /* bad argument (null, size mismatch, index out of range)        */
    MIRT_E_HANDLE = -2,     /* unknown or already released handle                            */
    MIRT_E_NAME = -3,       /* no kernel of that name (WebCL: INVALID_KERNEL_NAME)           */
    MIRT_E_UNSET = -4,      /* enqueue with an argument never set (INVALID_KERNEL_ARGS)      */
    MIRT_E_RANGE = -5,      /* buffer too small for what the launch would touch              */
    MIRT_E_DEVICE = -6,     /* HIP runtime error (message has the hipError string)           */
    MIRT_E_NODEVICE = -7,   /* no gfx950 device / extension not usable                       */
    MIRT_E_DATA = -8        /* device data failed validation (cell offsets not monotone ...) */
} mirt_status;

/* WebCL memory flags (A10 code.js:1083, 1175, 1312: MEM_READ_WRITE / MEM_READ_ONLY / MEM_WRITE_ONLY) */
#define MIRT_MEM_READ_WRITE 1u
#define MIRT_MEM_WRITE_ONLY 2u
#define MIRT_MEM_READ_ONLY 4u

/* ---- platform / device: webcl.getPlatforms(), platform.getDevices(), device.getInfo()
 *      (A10 code.js:483-498, 623-631) ------------------------------------------------- */
MIRT_API int mirt_device_count(void);
MIRT_API int mirt_device_name(int device, char* out, size_t cap);
MIRT_API const char* mirt_version(void);
/* Bumped whenever an existing entry point changes its parameters or the meaning of a value (a caller built against an older header would link and
 * pass shifted arguments).  4: mirt_gather takes n_tiles and a transport enum (round 3); mirt_pass_desc.acu may be NULL (round 4).  Check
 * mirt_abi_version() == MIRT_ABI_VERSION before anything else. */
#define MIRT_ABI_VERSION 4
MIRT_API int mirt_abi_version(void);

/* ---- context + command queue: webcl.createContext(device) + ctx.createCommandQueue()
 *      (A10 code.js:582, 592); release() of both (code.js:1539-1552) ------------------- */
MIRT_API int mirt_ctx_create(int device, mirt_ctx** out);
MIRT_API int mirt_ctx_destroy(mirt_ctx* ctx);
/* last error text of `ctx` (or of the calling thread when ctx is NULL); never NULL */
MIRT_API const char* mirt_last_error(mirt_ctx* ctx);
/* run on a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own */
MIRT_API int mirt_ctx_set_stream(mirt_ctx* ctx, void* hip_stream);
/* queue.finish() (A10 code.js:1096, 1406, 1533) */
MIRT_API int mirt_finish(mirt_ctx* ctx);

/* ---- buffers: ctx.createBuffer(flags, bytes) (A10 code.js:1083, 1117-1118, 1149, 1175-1177,
 *      1221-1224, 1268-1270, 1312, 1375, 1428), buffer.release() ----------------------- */
MIRT_API int mirt_buf_create(mirt_ctx* ctx, size_t bytes, unsigned flags, mirt_buf** out);
/* adopt device memory owned by the caller (a torch tensor's data_ptr); release() does not free it.  The runtime cannot see writes
 * made to such memory behind its back, so nothing derived from its contents is cached: cell-offset tables are re-validated and
 * triangles re-prepared on every launch that uses a wrapped buffer as geometry. */
MIRT_API int mirt_buf_wrap(mirt_ctx* ctx, void* device_ptr, size_t bytes, mirt_buf** out);
/* tell the runtime that a buffer's contents were changed outside mirt_buf_write / the kernels (through mirt_buf_device_ptr, or by
 * another stream): drops the cached validation / preparation of that buffer and invalidates recordings that relied on it */
MIRT_API int mirt_buf_invalidate(mirt_buf* buf);
MIRT_API int mirt_buf_release(mirt_buf* buf);
MIRT_API size_t mirt_buf_size(const mirt_buf* buf);
MIRT_API void* mirt_buf_device_ptr(const mirt_buf* buf);
/* queue.enqueueWriteBuffer(buf, blocking, offset, nbytes, typedArray, []) (A10 code.js:1153, 1183-1185, ...) */
MIRT_API int mirt_buf_write(mirt_buf* buf, size_t offset, size_t nbytes, const void* host, int blocking);
/* queue.enqueueReadBuffer(buf, blocking, offset, nbytes, typedArray, []) (A10 code.js:1070, 1532) */
MIRT_API int mirt_buf_read(mirt_buf* buf, size_t offset, size_t nbytes, void* host, int blocking);

/* ---- program + kernels: ctx.createProgram(src); program.build(); program.createKernel(name)
 *      (A10 code.js:596-607, 1048, 1087, 1107, 1158, 1206, 1256, 1307, 1348, 1366, 1422, 1446,
 *      1463, 1483).  There is no JIT: `mirt_program_check` scans the OpenCL C text the host
 *      would have compiled and reports, for every `__kernel void NAME(`, whether a built-in
 *      HIP kernel of that name exists; `missing` receives a comma-separated list of those
 *      that do not (the WebCL build log).  Returns the number of missing kernels, or <0. ---- */
MIRT_API int mirt_program_check(mirt_ctx* ctx, const char* source, char* missing, size_t cap);
/* Which assignment's kernel set an OpenCL C text asks for: 10 (Assign10), 7, 4, 1, or 0 = not built (A02/03/05/06/08/09).
 * The kernel NAMES collide across assignments (initTrace, meshTrace), so the sets other than A10's are reached through
 * mirt_kernel_get with a dialect prefix: "A07:meshTrace", "A04:initTrace", "A01:raytrace", ... */
MIRT_API int mirt_program_dialect(const char* source);
/* names: sizeofRay sizeofPoi initAcu initTrace sphereTrace triangleTrace meshTrace lightRender
 * initShadowTrace sphereShadowTrace triangleShadowTrace sceneRender bouncePaths copyToPixel   (A10, A10 code.cl:440-1386)
 * A07:sizeofRay A07:initTrace A07:meshTrace  (A07 code.cl:307-335, 475-626)   A04:sizeofRay A04:initTrace A04:meshTrace
 * (A04 code.cl:200-215, 262-315)   A01:raytrace (A01 code.cl:116-147) */
MIRT_API int mirt_kernel_get(mirt_ctx* ctx, const char* name, mirt_kernel** out);
MIRT_API int mirt_kernel_release(mirt_kernel* k);
/* number of arguments of the kernel (WebCL kernel.getInfo(KERNEL_NUM_ARGS)) */
MIRT_API int mirt_kernel_num_args(const mirt_kernel* k);
/* kernel.setArg(i, typedArray): scalars are 4 bytes, float16 64 bytes, AABB 32 bytes in the
 * host packing (min,1,max,1) (A10 code.js:610-621, 1089, 1127-1131, ...).  Size must match. */
MIRT_API int mirt_kernel_set_arg(mirt_kernel* k, unsigned index, size_t size, const void* value);
/* kernel.setArg(i, webclBuffer) */
MIRT_API int mirt_kernel_set_arg_buf(mirt_kernel* k, unsigned index, mirt_buf* buf);
/* kernel.getWorkGroupInfo(device, KERNEL_PREFERRED_WORK_GROUP_SIZE_MULTIPLE) (A10 code.js:656): 64 */
MIRT_API int mirt_kernel_preferred_multiple(const mirt_kernel* k);
/* queue.enqueueNDRangeKernel(kernel, dim, null, globalWS, localWS) (A10 code.js:1095, 1302, 1330,
 * 1339, 1343, 1399, 1405, 1414, 1503, 1507, 1511, 1519, 1527).  `local` may be NULL. */
MIRT_API int mirt_enqueue(mirt_ctx* ctx, mirt_kernel* k, unsigned dim, const size_t* global, const size_t* local);

/* ---- extension: the whole pass in one launch -------------------------------------------
 * One call == one executeRender() of the reference minus the read-back (A10 code.js:1806-1854):
 * initTrace, closest hits, lightRender, per-light shadow + shade, `bounces` bounce segments,
 * accumulated into `acu`; optionally followed by copyToPixel.  Results are bit-identical to
 * enqueueing the fourteen kernels one by one.  Rows [row0, row0+nrows) of the image are
 * rendered; seeds/acu/pixel/radiance are tile-local (nrows*width[*rpp] elements), ray ids
 * stay global so a frame does not depend on how it is tiled over GPUs. */
#define MIRT_MAX_LIGHTS 8
#define MIRT_MAX_MESHES 16

typedef struct mirt_grid {          /* one cell-sorted primitive set as the host uploads it        */
    mirt_buf* prims;                /* spheres: float4 (c, r^2) | triangles: 3 x float4 positions  */
    mirt_buf* normals;              /* triangles: 3 x float4 normals; NULL for spheres             */
    mirt_buf* matid;                /* uint per primitive; NULL for a mesh (uses mesh_matid)       */
    mirt_buf* cell_offsets;         /* uint[n^3 + 1]                                               */
    float bounds[8];                /* (min,1,max,1), bounds2AABB, A10 code.js:610-621             */
    uint32_t n_slabs;
    uint32_t mesh_matid;
} mirt_grid;

typedef struct mirt_light {         /* Light.to{Shadow,SceneRender,LightRender}Info, A10 code.js:323-352 */
    float shadow[16];
    float scene[16];
    float light[16];
} mirt_light;

typedef struct mirt_pass_desc {
    uint32_t struct_size;           /* sizeof(mirt_pass_desc), for ABI evolution                   */
    uint32_t width, height, rays_per_pixel;
    uint32_t row0, nrows;           /* row tile; nrows == 0 means the whole image.  A tile holds at most 2^32 - 256 rays
                                     * (nrows * width * rays_per_pixel: MIRT_E_ARG beyond; cut the rows over more calls) */
    uint32_t bounces;               /* 5 == reference (A10 code.js:1829)                           */
    uint32_t pass_index;            /* 1-based `passes` counter (A10 code.js:1850)                 */
    float cam[16];                  /* Camera.toFloat32Array, A10 code.js:250-258                  */
    float scene_bounds[8];
    float focal_length, lens_rad;   /* A10 code.js:1129-1130 (lens_rad = lens_diameter/2)          */
    uint32_t n_lights, n_meshes;
    const mirt_grid* spheres;       /* NULL: none                                                  */
    const mirt_grid* triangles;     /* NULL: none                                                  */
    const mirt_grid* meshes;        /* [n_meshes]                                                  */
    const mirt_light* lights;       /* [n_lights]                                                  */
    mirt_buf* material;             /* float4 per material (splitMaterialData, code.js:1774-1782)  */
    mirt_buf* seeds;                /* int32 per local ray, read-modify-write                      */
    mirt_buf* acu;                  /* float4 per local ray, accumulated into (zero it first).  May be NULL for a frame's first pass
                                     * (mirt_render_first_pass) that resolves its pixels itself: see below                       */
    mirt_buf* pixel;                /* optional: uchar4 per local pixel, written by copyToPixel    */
    mirt_buf* radiance;             /* optional: float4 per local pixel, un-scaled sequential sums */
} mirt_pass_desc;

/* rays_per_pixel must be k*k, as the reference host makes it (A10 code.js:540): MIRT_E_ARG otherwise */
MIRT_API int mirt_render_pass(mirt_ctx* ctx, const mirt_pass_desc* desc);
/* The first pass of a frame with preRender's initAcu (A10 code.js:1078-1099, code.cl:448-456) folded in: `acu` is not read, every
 * accumulator starts at (0,0,0,0).  Saves the zeroing launch and half of the pass's memory traffic; results equal
 * mirt_zero(acu) + mirt_render_pass.
 * copyToPixel INSIDE the pass: when a pixel or radiance buffer is given and rays_per_pixel divides 256 (1, 4, 16, 64, 256: a block of
 * 256 consecutive ray ids then holds whole pixels), every block sums its pixels' accumulators in LDS in copyToPixel's order (A10
 * code.cl:1366-1386: sequential in i, from +0) and writes pixel / radiance itself -- no second kernel, no re-read of the accumulators.
 * Then, and only then, `acu` may be NULL: nothing per ray but the seed touches memory (8 B per sample + 20 B per pixel) and the
 * 16 bytes per ray are never allocated; with `acu` given it is written as before (what a second progressive pass needs).  Results
 * are bit-identical either way.  In this mode the optimistic / exact kernel pair hands over whole blocks of 256 samples
 * (mirt_pass_deferred counts them as such).  MIRT_INPASS_RESOLVE=0 in the environment keeps the separate copyToPixel.
 * A LATER pass (mirt_render_pass) with a pixel or radiance buffer resolves inside the pass under the same condition -- `acu` is then read and
 * written as ever, and the separate copyToPixel's second read of it is saved; the runtime's command-stream fusion (mirt_ctx_set_fusion) folds the
 * host's recorded copyToPixel into the pass the same way, with the factor the host passed.
 * MORE than 256 rays per pixel, 256 times a power of two up to 32 (the squares among them: 1024 = BASELINE config 5's 32 x 32 lens grid,
 * and 4096): a pixel spans 4 (16) blocks, so the pass is queued as 4 (16) launches, launch c rendering the c-th block of every pixel and
 * going on from the sums launch c - 1 left in `radiance` (or in the context's scratch buffer when the caller passes none): the
 * reference's one chain of additions, cut at multiples of 256 and carried through memory at 16 B per pixel and launch instead of 16 B
 * per ray.  `acu` may be NULL there too; results are bit-identical. */
MIRT_API int mirt_render_first_pass(mirt_ctx* ctx, const mirt_pass_desc* desc);
/* Two ways to run the pass, identical results.  Default: the optimistic pair -- a kernel whose divisions are 3-operation
 * forms proven bit-exact inside a guard window (exhaustively, on the device: profiles/r1_divcheck_exhaustive.txt), plus the
 * exact kernel re-running the samples whose rays left the window (NaN rays, axis-parallel directions, ...); it needs every
 * grid to pass the geometry-side window check, otherwise the pass silently is the exact one.  mirt_ctx_set_exact_only(ctx, 1):
 * one kernel whose every division is the compiler's correctly rounded expansion.  The pair is queued without a host round trip
 * (the exact kernel walks the optimistic kernel's bit mask on the device).  mirt_pass_deferred: how many samples the last pass
 * re-ran through the exact kernel -- counted when asked (one small launch + a stream sync), valid until the next pass. */
MIRT_API int mirt_pass_deferred(mirt_ctx* ctx, uint64_t* samples);
MIRT_API int mirt_ctx_set_exact_only(mirt_ctx* ctx, int on);

/* Command-stream fusion: the reference host's pass, kernel by kernel, at the fused pass's speed -- with no change to the host.
 * executeRender (A10 code.js:1806-1854) issues a pass as 44+ enqueues whose every stage round-trips Ray / Poi / shadow Ray / acu
 * through HBM (4.2 KB per sample; the fused pass moves 24 B).  At level 2 the runtime holds back the enqueues of the Assign10 pass
 * kernels from an initTrace on, and when the stream up to the copyToPixel IS executeRender's sequence over one consistent set of
 * buffers and arguments -- initTrace; sphere / triangle / mesh Trace; lightRender per light; per light {initShadowTrace, the
 * any-hit kernel of every set, sceneRender}; any number of {bouncePaths, closest-hit kernels, that per-light block}; copyToPixel --
 * it runs as ONE launch of the fused pass + the recorded copyToPixel.  Anything else is launched enqueue by enqueue, in order,
 * exactly as at level 0: a different kernel order, mixed buffers or changed geometry arguments inside the pass, a global size
 * smaller than the ray count, and any command that observes or changes device state while enqueues are held (buffer read / write /
 * release, mirt_zero, mirt_seed_fill, mirt_render_pass, capture, timers, gather, destroy ...) flush the held stream first.
 * What level 2 trades, and why it is off by default in this ABI (the WebCL object model above it turns it on: end of this comment):
 *   - seeds, acu and pixel after the pass are bit-identical to level 0 (tests/test_fusion.py replays the reference host's own call
 *     stream both ways); the Ray, Poi and shadow-Ray buffers are NOT written by a fused pass -- they keep their previous contents.
 *     The reference host never reads them (it cannot: it does not know their layout beyond sizeof).
 *   - mirt_finish inside a held pass returns without draining anything (the reference calls finish() after every sceneRender,
 *     code.js:1406); the work runs at the copyToPixel.  Host-side timing of individual kernels is therefore meaningless.  A held pass
 *     that involves a wrapped buffer (mirt_buf_wrap: memory the caller can reach behind the ABI) IS drained by mirt_finish, and
 *     mirt_buf_device_ptr drains whatever is held.
 *   - errors of a held enqueue (a buffer too small, a grid failing validation) are reported by the call that flushes it, and the
 *     held enqueues after the failing one are dropped (at level 0 the host would have stopped at that enqueue's exception).
 * Level 0 (the default of mirt_ctx_create): every enqueue launches its kernel.  The environment variable MIRT_FUSION=0|2 sets the level
 * of every new context.  The WebCL object model above this ABI (host/webcl.js webcl.createContext) asks for level 2 itself unless the
 * variable is set: its one client, the reference page, cannot observe the difference.
 * mirt_ctx_fused_passes: how many passes of this context ran fused. */
MIRT_API int mirt_ctx_set_fusion(mirt_ctx* ctx, int level);
MIRT_API int mirt_ctx_fused_passes(mirt_ctx* ctx, uint64_t* count);

/* seeds[i] = 1 + (mix32((first_ray + i) ^ 0x9E3779B9 ^ seed_base) mod 2147483646): the
 * reproducible stand-in for the host's Math.random() seeding (A10 code.js:1140-1146). */
MIRT_API int mirt_seed_fill(mirt_ctx* ctx, mirt_buf* seeds, uint64_t first_ray, uint64_t count, uint32_t seed_base);
/* initAcu over a whole buffer (A10 code.js:1078-1099) */
MIRT_API int mirt_zero(mirt_ctx* ctx, mirt_buf* buf);

/* ---- uniform-grid build on the device: splitSphereData / splitTriangleData / splitMeshData (A10 code.js:1554-1772, 899-1041)
 * and Mesh.normalize/scale/translate (code.js:114-169) as a count / scan / stable-sort / gather pipeline in fp64, reproducing
 * the reference's cell order, per-cell input order and its dropped-on-the-max-face quirk bit for bit.  Every output is a
 * new buffer the caller owns (release it) and can bind to a kernel or put in a mirt_grid. ---------------------------------- */
typedef struct mirt_grid_build_desc {
    uint32_t struct_size;
    uint32_t kind;          /* 0: spheres, 4 doubles per primitive (cx, cy, cz, r); 1: triangles, 9 doubles (p0, p1, p2) */
    uint32_t count;         /* primitives */
    uint32_t n_slabs;       /* cells per axis */
    double bounds[6];       /* min x,y,z, max x,y,z of the set (Bounds, lib/utilities.js:389-422) */
    mirt_buf* prims_f64;    /* device buffer of doubles, uploaded with mirt_buf_write */
} mirt_grid_build_desc;
/* ---- mesh ingest on the device: parseMeshJSON (A10 tri/meshDataVersion1.js:12-78) for one (node, mesh) pair of an Assimp-style mesh
 * file: every triangle corner de-indexed, its position through the node's model matrix (gl-matrix vec3.transformMat4 on Float32Array
 * operands: fp32 matrix, double sums, fp32 store) and its normal through the normal matrix (vec3.transformMat3), written as the fp64
 * soups mirt_grid_build / mirt_grid_gather_triangles consume, at corner offset `first_corner`; `bounds6` (6 floats: min xyz, max xyz,
 * initialise to +inf / -inf) is merged with the mesh's TRANSFORMED vertices, all of them (:33-37).  The normal matrix
 * (mat3.normalFromMat4 of the model matrix, nine floats) is per node and computed by the caller.  Synchronises (an index past the
 * vertex array is MIRT_E_DATA).  Bit-identical to the reference host's arrays (tests/test_js_host.py, device vs host ingest). */
typedef struct mirt_mesh_ingest_desc {
    uint32_t struct_size;
    uint32_t n_vertices;        /* entries of positions / normals (3 doubles each) */
    uint32_t n_corners;         /* 3 x triangles of this mesh: indices.length, or n_vertices when un-indexed */
    uint32_t first_corner;      /* where this (node, mesh) pair's corners start in the output soups */
    float model[16];            /* node.modelMatrix narrowed to fp32 (mat4.copy into a Float32Array), column-major */
    float normal_mat[9];        /* mat3.normalFromMat4(model) */
    mirt_buf* positions_f64;    /* mesh.vertexPositions */
    mirt_buf* normals_f64;      /* mesh.vertexNormals */
    mirt_buf* indices_u32;      /* mesh.indices, or NULL */
} mirt_mesh_ingest_desc;
MIRT_API int mirt_mesh_ingest(mirt_ctx* ctx, const mirt_mesh_ingest_desc* d, mirt_buf* pos9_out, mirt_buf* nor9_out, mirt_buf* bounds6);

/* cell_offsets: uint[n^3+1]; order: uint[total], order[slot] = input index of the primitive in that slot */
MIRT_API int mirt_grid_build(mirt_ctx* ctx, const mirt_grid_build_desc* d, mirt_buf** cell_offsets, mirt_buf** order, uint32_t* total);
/* slot arrays from `order`.  Triangles: 3 x float4 per slot (w = pad_w: 0 in A07/A10, 1 for A04's positions), after up to four
 * fp64 per-axis steps (op 0 subtract, 1 multiply, 2 add; vecs = 3 doubles per step) applied in order, then narrowed to fp32.
 * nor_f64 / nor_out may be NULL. */
MIRT_API int mirt_grid_gather_triangles(mirt_ctx* ctx, mirt_buf* order, uint32_t total, mirt_buf* pos_f64, mirt_buf* nor_f64,
                                        uint32_t nsteps, const int32_t* ops, const double* vecs, float pad_w,
                                        mirt_buf** pos_out, mirt_buf** nor_out);
MIRT_API int mirt_grid_gather_spheres(mirt_ctx* ctx, mirt_buf* order, uint32_t total, mirt_buf* sph_f64, mirt_buf** out);   /* float4 (c, r*r) */
MIRT_API int mirt_grid_gather_u32(mirt_ctx* ctx, mirt_buf* order, uint32_t total, mirt_buf* in_u32, mirt_buf** out);       /* material ids */

/* ---- diagnostics: evaluate one primitive of the numerics contract element-wise on the device
 * (op: 0 a/b, 1 sqrt, 2 sin, 3 cos, 4 getRand(seed=bits of a), 5 next LCG state, 6 min, 7 max, 8 fmin, 9 fmax,
 * 10 normalize(a,b,1).x, 11/12 concentric_distort(a,b).x/.y, 13 (int)a, 14 (uint)a, 25 clamp(a,0,b);
 * float3 arguments as three consecutive floats per element: 20 dot, 21 cross (3 outputs), 22 length, 23 distance,
 * 24 normalize (3 outputs), 26 mad(a.x,a.y,a.z), 27 the four contraction shapes a*b+c, a*b-c, c-a*b, a*b+c*a (4 outputs)).
 * Lets a test compare device bits with AMD's OpenCL library (oracle/probe/builtins.cl) over millions of inputs. ---- */
MIRT_API int mirt_debug_numerics(mirt_ctx* ctx, int op, mirt_buf* a, mirt_buf* b, mirt_buf* out, size_t n);

/* the PREPARED copy the runtime keeps of a triangle position buffer holding `count` triangles (3 x float4 each), built on first use by a pass
 * or frame kernel and rebuilt when the buffer's contents change: `count` records of 48 bytes {p0, n.x}{e1, n.y}{e2, n.z}, one bounding sphere
 * (float4) per 16 records, and -- for at most 96 records -- the candidate sweep's plane list (64-byte aligned; csrc/pt_launch.hpp).  Copies up
 * to `bytes` of it to `out` and reports its size in *total.  For the tests that pin its layout against the CPU restatement. */
MIRT_API int mirt_debug_prepared(mirt_ctx* ctx, mirt_buf* positions, uint32_t count, void* out, size_t bytes, size_t* total);

/* counts mismatches between the shared-reciprocal division forms of pt_numerics.hpp and the compiler's correctly
 * rounded division over `count` generated (n, d) pairs; `out16` receives 16 uint64 (see k_divCheck).
 * mode 0/1/2: random pairs inside the windows; 3: all 2^32 denominators of the reciprocal; 4: every numerator mantissa
 * against the `count` denominator mantissas starting at `seed` (2^23 launches' worth covers all 2^46 pairs); 5: the 9-operation
 * correctly rounded sqrt over all 2^32 bit patterns (out[1] mismatches of cl_sqrt, out[2] / out[3] of the bare core / outside denormals) */
MIRT_API int mirt_debug_divcheck(mirt_ctx* ctx, int mode, uint64_t seed, uint64_t count, mirt_buf* out16);

/* ---- launch-bound sequences as HIP graphs ------------------------------------------------------
 * The reference's executeRender() is 44+ enqueues per pass (A10 code.js:1806-1854); on a 320x240 canvas at one ray per pixel (the
 * page's defaults, index.html:46, code.js:400) every one of them is a few microseconds of work behind a launch.  Between
 * mirt_capture_begin and mirt_capture_end every mirt_enqueue / mirt_render_pass / mirt_zero / mirt_seed_fill on the context is
 * recorded instead of run; mirt_graph_launch replays the recording with the argument values it was recorded with.  Run the sequence
 * once normally first: whatever needs a host round trip (cell-table validation, triangle preparation, scratch growth) is cached
 * by that run and refused (MIRT_E_ARG) inside a capture, as are mirt_finish, buffer reads / writes and timers.
 * A recording holds raw device pointers.  Every allocation it touched is pinned by identity: mirt_graph_launch returns MIRT_E_HANDLE
 * once one of those buffers has been released, once the context's scratch memory has been reallocated, or once geometry the
 * recording was validated against (cell offsets, triangle positions) has been rewritten -- record the sequence again. */
typedef struct mirt_graph mirt_graph;
MIRT_API int mirt_capture_begin(mirt_ctx* ctx);
MIRT_API int mirt_capture_end(mirt_ctx* ctx, mirt_graph** out);
MIRT_API int mirt_graph_launch(mirt_ctx* ctx, mirt_graph* graph);
MIRT_API int mirt_graph_release(mirt_graph* graph);

/* ---- several devices in one process: row tiles + the one exchange of the path --------------------------------------------
 * The reference is a single-device page (one context, one queue: A10 code.js:582, 592).  Every ray is independent, so a frame
 * shards by pixel rows; ray ids stay global (mirt_pass_desc.row0 / nrows), which makes the frame independent of the tiling.
 * A group is N contexts, one per device, driven from the one host thread: enqueue the tile passes on each context (launches are
 * asynchronous), then mirt_gather assembles the tiles' buffers on the root device.  Transports: RCCL (ncclCommInitAll + one grouped
 * ncclSend / ncclRecv exchange: N - 1 peers, N - 1 distinct xGMI links into the root; librccl is loaded on first use), or plain copies
 * queued on the root's stream (hipMemcpyPeerAsync across devices, a device copy inside one), each ordered after the tile's own stream.
 * MIRT_GATHER_AUTO picks RCCL for N > 1 distinct devices when librccl loads, copies otherwise; _RCCL / _COPY force one (a one-GPU box
 * exercises the RCCL calls with a one-rank communicator).
 * Rehearsal switch: with MIRT_GROUP_ALLOW_REPEATED_DEVICES=1 in the environment mirt_group_create accepts a device listed several
 * times -- N contexts on the devices at hand, so the whole N-tile path runs on a one-GPU box; such a group gathers by copies only. */
typedef struct mirt_group mirt_group;
MIRT_API int mirt_group_create(const int* device_ids, int n, mirt_group** out);
MIRT_API int mirt_group_size(const mirt_group* g);
MIRT_API mirt_ctx* mirt_group_ctx(const mirt_group* g, int index);     /* owned by the group: do not mirt_ctx_destroy it */
MIRT_API int mirt_group_destroy(mirt_group* g);                       /* destroys its contexts and everything created on them */
MIRT_API int mirt_group_finish(mirt_group* g);                        /* queue.finish() on every device */
/* contiguous row tiles whose sizes differ by at most one row (the first height % n_tiles tiles take the extra one) */
MIRT_API void mirt_tile_rows(uint32_t height, uint32_t n_tiles, uint32_t index, uint32_t* row0, uint32_t* nrows);
/* out[sum of tile_bytes[0..i-1] ...] = the first tile_bytes[i] bytes of tiles[i] (a buffer of context i), for every i < n_tiles
 * (n_tiles must equal mirt_group_size); `out` is a buffer of context `root`.  Ordered after the work queued on each context; complete after mirt_finish(root ctx). */
enum { MIRT_GATHER_AUTO = 0, MIRT_GATHER_RCCL = 1, MIRT_GATHER_COPY = 2 };
MIRT_API int mirt_gather(mirt_group* g, mirt_buf* const* tiles, const size_t* tile_bytes, int n_tiles, mirt_buf* out, int root, int transport);
/* What a group can do and what the last gather did.  mirt_group_create asks hipDeviceCanAccessPeer for every pair of distinct devices and enables peer
 * access both ways; mirt_group_peer_access(g, i, j) = 1 when context i's device reads context j's device's memory directly (always 1 for i == j and for
 * contexts that share a device), 0 when the runtime refused -- then a copy between them is staged through host memory by the HIP runtime.
 * mirt_gather_route(g, tile): how the LAST mirt_gather moved that tile -- MIRT_GATHER_AUTO may pick copies without saying so (librccl absent, a
 * rehearsal group), and a staged copy is an order of magnitude slower than a peer one: a first run on N real GPUs is diagnosable from these. */
enum { MIRT_ROUTE_NONE = 0,   /* no gather yet, or an empty tile                                       */
       MIRT_ROUTE_RCCL = 1,   /* ncclSend / ncclRecv                                                    */
       MIRT_ROUTE_PEER = 2,   /* hipMemcpyPeerAsync between devices with peer access enabled (xGMI P2P) */
       MIRT_ROUTE_STAGED = 3, /* hipMemcpyPeerAsync WITHOUT peer access: bounced through host memory    */
       MIRT_ROUTE_LOCAL = 4 };/* the tile already lives on the root's device: a device-local copy      */
MIRT_API int mirt_group_peer_access(const mirt_group* g, int i, int j);
MIRT_API int mirt_gather_route(const mirt_group* g, int tile);

/* ---- measurement: HIP events on the context's stream ---------------------------------- */
MIRT_API int mirt_timer_start(mirt_ctx* ctx);
MIRT_API int mirt_timer_stop_ms(mirt_ctx* ctx, float* ms);   /* synchronises */
/* per-kernel events inside mirt_render_pass: duration of the fused pass kernel and of the resolve
 * (copyToPixel) kernel of the most recent profiled pass, on the stream they were launched on */
MIRT_API int mirt_ctx_set_profiling(mirt_ctx* ctx, int on);
MIRT_API int mirt_pass_timing(mirt_ctx* ctx, float* fused_ms, float* resolve_ms);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
