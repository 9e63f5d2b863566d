// mirt_napi.cc -- N-API addon (raw node_api.h, N-API v3+; Node 12 has no napi.h wrapper here).
//
// A thin, flat binding of include/mirt.h for the JavaScript host: every export is one
// mirt_* call.  Handles travel as napi_external values.  Errors become JS exceptions whose
// `.code` is the mirt_status name and whose message is mirt_last_error().  The WebCL-shaped
// object model the reference host drives (webcl.getPlatforms() ... kernel.setArg ...) is
// built on top of these in host/webcl.js, in JavaScript.
#include <node_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/mirt.h"

namespace {

const char* status_name(int rc) {
    switch (rc) {
        case MIRT_E_ARG: return "MIRT_E_ARG";
        case MIRT_E_HANDLE: return "MIRT_E_HANDLE";
        case MIRT_E_NAME: return "MIRT_E_NAME";
        case MIRT_E_UNSET: return "MIRT_E_UNSET";
        case MIRT_E_RANGE: return "MIRT_E_RANGE";
        case MIRT_E_DEVICE: return "MIRT_E_DEVICE";
        case MIRT_E_NODEVICE: return "MIRT_E_NODEVICE";
        case MIRT_E_DATA: return "MIRT_E_DATA";
        default: return "MIRT_E_UNKNOWN";
    }
}

napi_value throw_mirt(napi_env env, int rc, mirt_ctx* ctx) {
    napi_throw_error(env, status_name(rc), mirt_last_error(ctx));
    return nullptr;
}
napi_value throw_type(napi_env env, const char* msg) {
    napi_throw_type_error(env, "MIRT_E_ARG", msg);
    return nullptr;
}

#define ARGS(n)                                                         \
    size_t argc = (n);                                                  \
    napi_value argv[(n) > 0 ? (n) : 1];                                 \
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok) return throw_type(env, "bad call"); \
    if (argc < (size_t)(n)) return throw_type(env, "too few arguments")

bool get_ext(napi_env env, napi_value v, void** out) {
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_external) return false;
    return napi_get_value_external(env, v, out) == napi_ok;
}
bool get_u32(napi_env env, napi_value v, uint32_t* out) { return napi_get_value_uint32(env, v, out) == napi_ok; }
bool get_f64(napi_env env, napi_value v, double* out) { return napi_get_value_double(env, v, out) == napi_ok; }
bool get_bytes(napi_env env, napi_value v, void** data, size_t* nbytes) {
    bool is_ta = false;
    if (napi_is_typedarray(env, v, &is_ta) != napi_ok) return false;
    if (is_ta) {
        napi_typedarray_type t;
        size_t len;
        napi_value ab;
        size_t off;
        if (napi_get_typedarray_info(env, v, &t, &len, data, &ab, &off) != napi_ok) return false;
        size_t es = 1;
        switch (t) {
            case napi_int16_array: case napi_uint16_array: es = 2; break;
            case napi_int32_array: case napi_uint32_array: case napi_float32_array: es = 4; break;
            case napi_float64_array: es = 8; break;
            default: es = 1; break;
        }
        *nbytes = len * es;
        return true;
    }
    bool is_ab = false;
    if (napi_is_arraybuffer(env, v, &is_ab) == napi_ok && is_ab) return napi_get_arraybuffer_info(env, v, data, nbytes) == napi_ok;
    return false;
}
napi_value mk_ext(napi_env env, void* p) { napi_value v; napi_create_external(env, p, nullptr, nullptr, &v); return v; }
napi_value mk_num(napi_env env, double d) { napi_value v; napi_create_double(env, d, &v); return v; }
napi_value undef(napi_env env) { napi_value v; napi_get_undefined(env, &v); return v; }

bool prop(napi_env env, napi_value obj, const char* name, napi_value* out) {
    bool has = false;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return false;
    if (napi_get_named_property(env, obj, name, out) != napi_ok) return false;
    napi_valuetype t;
    napi_typeof(env, *out, &t);
    return t != napi_undefined && t != napi_null;
}
bool prop_u32(napi_env env, napi_value obj, const char* name, uint32_t* out) { napi_value v; return prop(env, obj, name, &v) && get_u32(env, v, out); }
bool prop_f32(napi_env env, napi_value obj, const char* name, float* out) { napi_value v; double d; if (!prop(env, obj, name, &v) || !get_f64(env, v, &d)) return false; *out = (float)d; return true; }
bool prop_floats(napi_env env, napi_value obj, const char* name, float* dst, size_t n) {
    napi_value v; void* data; size_t nb;
    if (!prop(env, obj, name, &v) || !get_bytes(env, v, &data, &nb) || nb < n * 4) return false;
    memcpy(dst, data, n * 4);
    return true;
}
mirt_buf* prop_buf(napi_env env, napi_value obj, const char* name) { napi_value v; void* p = nullptr; if (prop(env, obj, name, &v) && get_ext(env, v, &p)) return (mirt_buf*)p; return nullptr; }

// ---- exports --------------------------------------------------------------------------
napi_value DeviceCount(napi_env env, napi_callback_info) { return mk_num(env, mirt_device_count()); }

napi_value DeviceName(napi_env env, napi_callback_info info) {
    ARGS(1);
    uint32_t d;
    if (!get_u32(env, argv[0], &d)) return throw_type(env, "deviceName(index)");
    char buf[256];
    int rc = mirt_device_name((int)d, buf, sizeof buf);
    if (rc) return throw_mirt(env, rc, nullptr);
    napi_value s;
    napi_create_string_utf8(env, buf, NAPI_AUTO_LENGTH, &s);
    return s;
}

napi_value Version(napi_env env, napi_callback_info) { napi_value s; napi_create_string_utf8(env, mirt_version(), NAPI_AUTO_LENGTH, &s); return s; }

napi_value CtxCreate(napi_env env, napi_callback_info info) {
    ARGS(1);
    uint32_t d;
    if (!get_u32(env, argv[0], &d)) return throw_type(env, "ctxCreate(device)");
    mirt_ctx* c = nullptr;
    int rc = mirt_ctx_create((int)d, &c);
    if (rc) return throw_mirt(env, rc, nullptr);
    return mk_ext(env, c);
}
napi_value CtxDestroy(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* c;
    if (!get_ext(env, argv[0], &c)) return throw_type(env, "ctxDestroy(ctx)");
    int rc = mirt_ctx_destroy((mirt_ctx*)c);
    if (rc) return throw_mirt(env, rc, nullptr);
    return undef(env);
}
napi_value Finish(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* c;
    if (!get_ext(env, argv[0], &c)) return throw_type(env, "finish(ctx)");
    int rc = mirt_finish((mirt_ctx*)c);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return undef(env);
}
napi_value CtxSetFusion(napi_env env, napi_callback_info info) {
    ARGS(2);
    void* c; uint32_t level;
    if (!get_ext(env, argv[0], &c) || !get_u32(env, argv[1], &level)) return throw_type(env, "ctxSetFusion(ctx, level)");
    int rc = mirt_ctx_set_fusion((mirt_ctx*)c, (int)level);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return undef(env);
}
napi_value CtxFusedPasses(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* c;
    if (!get_ext(env, argv[0], &c)) return throw_type(env, "ctxFusedPasses(ctx)");
    uint64_t n = 0;
    int rc = mirt_ctx_fused_passes((mirt_ctx*)c, &n);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return mk_num(env, (double)n);
}
napi_value BufCreate(napi_env env, napi_callback_info info) {
    ARGS(3);
    void* c; double bytes; uint32_t flags;
    if (!get_ext(env, argv[0], &c) || !get_f64(env, argv[1], &bytes) || !get_u32(env, argv[2], &flags)) return throw_type(env, "bufCreate(ctx, bytes, flags)");
    mirt_buf* b = nullptr;
    int rc = mirt_buf_create((mirt_ctx*)c, bytes < 0 ? 0 : (size_t)bytes, flags, &b);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return mk_ext(env, b);
}
napi_value BufRelease(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* b;
    if (!get_ext(env, argv[0], &b)) return throw_type(env, "bufRelease(buf)");
    int rc = mirt_buf_release((mirt_buf*)b);
    if (rc) return throw_mirt(env, rc, nullptr);
    return undef(env);
}
napi_value BufSize(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* b;
    if (!get_ext(env, argv[0], &b)) return throw_type(env, "bufSize(buf)");
    return mk_num(env, (double)mirt_buf_size((mirt_buf*)b));
}
napi_value BufRW(napi_env env, napi_callback_info info, bool write) {
    ARGS(4);
    void* b; double off, n; void* data; size_t nb;
    if (!get_ext(env, argv[0], &b) || !get_f64(env, argv[1], &off) || !get_f64(env, argv[2], &n) || !get_bytes(env, argv[3], &data, &nb))
        return throw_type(env, "buf{Write,Read}(buf, offset, nbytes, typedArray)");
    if (off < 0 || n < 0 || (size_t)n > nb) return throw_type(env, "nbytes exceeds the typed array");
    int rc = write ? mirt_buf_write((mirt_buf*)b, (size_t)off, (size_t)n, data, 0) : mirt_buf_read((mirt_buf*)b, (size_t)off, (size_t)n, data, 1);
    if (rc) return throw_mirt(env, rc, nullptr);
    return undef(env);
}
napi_value BufWrite(napi_env env, napi_callback_info info) { return BufRW(env, info, true); }
napi_value BufRead(napi_env env, napi_callback_info info) { return BufRW(env, info, false); }

napi_value ProgramCheck(napi_env env, napi_callback_info info) {
    ARGS(2);
    void* c;
    if (!get_ext(env, argv[0], &c)) return throw_type(env, "programCheck(ctx, source)");
    size_t len = 0;
    if (napi_get_value_string_utf8(env, argv[1], nullptr, 0, &len) != napi_ok) return throw_type(env, "programCheck: source must be a string");
    std::string src(len + 1, '\0');
    napi_get_value_string_utf8(env, argv[1], &src[0], len + 1, &len);
    char missing[2048];
    int n = mirt_program_check((mirt_ctx*)c, src.c_str(), missing, sizeof missing);
    if (n < 0) return throw_mirt(env, n, (mirt_ctx*)c);
    napi_value o, s;
    napi_create_object(env, &o);
    napi_set_named_property(env, o, "missing", mk_num(env, n));
    napi_create_string_utf8(env, missing, NAPI_AUTO_LENGTH, &s);
    napi_set_named_property(env, o, "log", s);
    return o;
}

napi_value ProgramDialect(napi_env env, napi_callback_info info) {
    ARGS(1);
    size_t len = 0;
    if (napi_get_value_string_utf8(env, argv[0], nullptr, 0, &len) != napi_ok) return throw_type(env, "programDialect(source)");
    std::string src(len + 1, '\0');
    napi_get_value_string_utf8(env, argv[0], &src[0], len + 1, &len);
    return mk_num(env, mirt_program_dialect(src.c_str()));
}

napi_value KernelGet(napi_env env, napi_callback_info info) {
    ARGS(2);
    void* c; char name[128]; size_t len;
    if (!get_ext(env, argv[0], &c) || napi_get_value_string_utf8(env, argv[1], name, sizeof name, &len) != napi_ok) return throw_type(env, "kernelGet(ctx, name)");
    mirt_kernel* k = nullptr;
    int rc = mirt_kernel_get((mirt_ctx*)c, name, &k);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return mk_ext(env, k);
}
napi_value KernelRelease(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* k;
    if (!get_ext(env, argv[0], &k)) return throw_type(env, "kernelRelease(kernel)");
    int rc = mirt_kernel_release((mirt_kernel*)k);
    if (rc) return throw_mirt(env, rc, nullptr);
    return undef(env);
}
napi_value KernelNumArgs(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* k;
    if (!get_ext(env, argv[0], &k)) return throw_type(env, "kernelNumArgs(kernel)");
    return mk_num(env, mirt_kernel_num_args((mirt_kernel*)k));
}
napi_value KernelPreferredMultiple(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* k;
    if (!get_ext(env, argv[0], &k)) return throw_type(env, "kernelPreferredMultiple(kernel)");
    return mk_num(env, mirt_kernel_preferred_multiple((mirt_kernel*)k));
}
// kernelSetArg(kernel, index, typedArray | bufferHandle)
napi_value KernelSetArg(napi_env env, napi_callback_info info) {
    ARGS(3);
    void* k; uint32_t idx;
    if (!get_ext(env, argv[0], &k) || !get_u32(env, argv[1], &idx)) return throw_type(env, "kernelSetArg(kernel, index, value)");
    void* p; size_t nb; int rc;
    if (get_ext(env, argv[2], &p)) rc = mirt_kernel_set_arg_buf((mirt_kernel*)k, idx, (mirt_buf*)p);
    else if (get_bytes(env, argv[2], &p, &nb)) rc = mirt_kernel_set_arg((mirt_kernel*)k, idx, nb, p);
    else return throw_type(env, "kernelSetArg: value must be a typed array or a buffer");
    if (rc) return throw_mirt(env, rc, nullptr);
    return undef(env);
}
bool get_sizes(napi_env env, napi_value arr, std::vector<size_t>* out) {
    bool is_arr = false;
    if (napi_is_array(env, arr, &is_arr) != napi_ok || !is_arr) return false;
    uint32_t n;
    napi_get_array_length(env, arr, &n);
    for (uint32_t i = 0; i < n; ++i) {
        napi_value e; double d;
        napi_get_element(env, arr, i, &e);
        if (!get_f64(env, e, &d) || d < 0) return false;
        out->push_back((size_t)d);
    }
    return true;
}
napi_value Enqueue(napi_env env, napi_callback_info info) {
    ARGS(5);
    void *c, *k; uint32_t dim;
    std::vector<size_t> g, l;
    if (!get_ext(env, argv[0], &c) || !get_ext(env, argv[1], &k) || !get_u32(env, argv[2], &dim) || !get_sizes(env, argv[3], &g))
        return throw_type(env, "enqueue(ctx, kernel, dim, globalWS[], localWS[]|null)");
    bool has_local = get_sizes(env, argv[4], &l);
    if (g.size() < dim || (has_local && l.size() < dim)) return throw_type(env, "enqueue: work-size arrays shorter than dim");
    int rc = mirt_enqueue((mirt_ctx*)c, (mirt_kernel*)k, dim, g.data(), has_local ? l.data() : nullptr);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return undef(env);
}

bool read_grid(napi_env env, napi_value o, mirt_grid* g) {
    memset(g, 0, sizeof *g);
    g->prims = prop_buf(env, o, "prims");
    g->normals = prop_buf(env, o, "normals");
    g->matid = prop_buf(env, o, "matid");
    g->cell_offsets = prop_buf(env, o, "cellOffsets");
    prop_u32(env, o, "meshMatId", &g->mesh_matid);
    return prop_floats(env, o, "bounds", g->bounds, 8) && prop_u32(env, o, "nSlabs", &g->n_slabs);
}
// renderPass(ctx, {width,height,raysPerPixel,row0,nrows,bounces,passIndex,cam,sceneBounds,focalLength,lensRad,
//                  spheres?,triangles?,meshes[],lights[{shadow,scene,light}],material,seeds,acu,pixel?,radiance?})
napi_value RenderPass(napi_env env, napi_callback_info info) {
    ARGS(2);
    void* c;
    if (!get_ext(env, argv[0], &c)) return throw_type(env, "renderPass(ctx, desc)");
    napi_value d = argv[1];
    mirt_pass_desc p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.bounces = 5;
    p.pass_index = 1;
    if (!prop_u32(env, d, "width", &p.width) || !prop_u32(env, d, "height", &p.height) || !prop_u32(env, d, "raysPerPixel", &p.rays_per_pixel))
        return throw_type(env, "renderPass: width/height/raysPerPixel");
    prop_u32(env, d, "row0", &p.row0);
    prop_u32(env, d, "nrows", &p.nrows);
    prop_u32(env, d, "bounces", &p.bounces);
    prop_u32(env, d, "passIndex", &p.pass_index);
    if (!prop_floats(env, d, "cam", p.cam, 16) || !prop_floats(env, d, "sceneBounds", p.scene_bounds, 8) ||
        !prop_f32(env, d, "focalLength", &p.focal_length) || !prop_f32(env, d, "lensRad", &p.lens_rad))
        return throw_type(env, "renderPass: cam (16 floats), sceneBounds (8 floats), focalLength, lensRad");
    mirt_grid sph, tri;
    std::vector<mirt_grid> meshes;
    std::vector<mirt_light> lights;
    napi_value v;
    if (prop(env, d, "spheres", &v)) { if (!read_grid(env, v, &sph)) return throw_type(env, "renderPass: spheres"); p.spheres = &sph; }
    if (prop(env, d, "triangles", &v)) { if (!read_grid(env, v, &tri)) return throw_type(env, "renderPass: triangles"); p.triangles = &tri; }
    if (prop(env, d, "meshes", &v)) {
        uint32_t n = 0;
        napi_get_array_length(env, v, &n);
        meshes.resize(n);
        for (uint32_t i = 0; i < n; ++i) { napi_value e; napi_get_element(env, v, i, &e); if (!read_grid(env, e, &meshes[i])) return throw_type(env, "renderPass: meshes[i]"); }
    }
    if (prop(env, d, "lights", &v)) {
        uint32_t n = 0;
        napi_get_array_length(env, v, &n);
        lights.resize(n);
        for (uint32_t i = 0; i < n; ++i) {
            napi_value e;
            napi_get_element(env, v, i, &e);
            if (!prop_floats(env, e, "shadow", lights[i].shadow, 16) || !prop_floats(env, e, "scene", lights[i].scene, 16) || !prop_floats(env, e, "light", lights[i].light, 16))
                return throw_type(env, "renderPass: lights[i] needs shadow/scene/light (16 floats each)");
        }
    }
    p.n_meshes = (uint32_t)meshes.size();
    p.meshes = meshes.data();
    p.n_lights = (uint32_t)lights.size();
    p.lights = lights.data();
    p.material = prop_buf(env, d, "material");
    p.seeds = prop_buf(env, d, "seeds");
    p.acu = prop_buf(env, d, "acu");
    p.pixel = prop_buf(env, d, "pixel");
    p.radiance = prop_buf(env, d, "radiance");
    bool first = false;   // desc.firstPass: initAcu folded into the pass (mirt_render_first_pass)
    { napi_value v; bool has = false; if (napi_has_named_property(env, d, "firstPass", &has) == napi_ok && has && napi_get_named_property(env, d, "firstPass", &v) == napi_ok) napi_get_value_bool(env, v, &first); }
    int rc = first ? mirt_render_first_pass((mirt_ctx*)c, &p) : mirt_render_pass((mirt_ctx*)c, &p);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return undef(env);
}

napi_value SeedFill(napi_env env, napi_callback_info info) {
    ARGS(5);
    void *c, *b; double first, count; uint32_t base;
    if (!get_ext(env, argv[0], &c) || !get_ext(env, argv[1], &b) || !get_f64(env, argv[2], &first) || !get_f64(env, argv[3], &count) || !get_u32(env, argv[4], &base))
        return throw_type(env, "seedFill(ctx, buf, firstRay, count, seedBase)");
    int rc = mirt_seed_fill((mirt_ctx*)c, (mirt_buf*)b, (uint64_t)first, (uint64_t)count, base);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return undef(env);
}
napi_value Zero(napi_env env, napi_callback_info info) {
    ARGS(2);
    void *c, *b;
    if (!get_ext(env, argv[0], &c) || !get_ext(env, argv[1], &b)) return throw_type(env, "zero(ctx, buf)");
    int rc = mirt_zero((mirt_ctx*)c, (mirt_buf*)b);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return undef(env);
}
// launch-bound sequences as HIP graphs (mirt.h): captureBegin(ctx), captureEnd(ctx) -> graph, graphLaunch(ctx, graph), graphRelease(graph)
napi_value CaptureBegin(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* c;
    if (!get_ext(env, argv[0], &c)) return throw_type(env, "captureBegin(ctx)");
    int rc = mirt_capture_begin((mirt_ctx*)c);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return undef(env);
}
napi_value CaptureEnd(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* c;
    if (!get_ext(env, argv[0], &c)) return throw_type(env, "captureEnd(ctx)");
    mirt_graph* g = nullptr;
    int rc = mirt_capture_end((mirt_ctx*)c, &g);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return mk_ext(env, g);
}
napi_value GraphLaunch(napi_env env, napi_callback_info info) {
    ARGS(2);
    void *c, *g;
    if (!get_ext(env, argv[0], &c) || !get_ext(env, argv[1], &g)) return throw_type(env, "graphLaunch(ctx, graph)");
    int rc = mirt_graph_launch((mirt_ctx*)c, (mirt_graph*)g);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return undef(env);
}
napi_value GraphRelease(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* g;
    if (!get_ext(env, argv[0], &g)) return throw_type(env, "graphRelease(graph)");
    int rc = mirt_graph_release((mirt_graph*)g);
    if (rc) return throw_mirt(env, rc, nullptr);
    return undef(env);
}
// gridBuild(ctx, kind, primsF64Buf, count, nSlabs, boundsFloat64Array(6)) -> {offsets, order, total}
napi_value GridBuild(napi_env env, napi_callback_info info) {
    ARGS(6);
    void *c, *pb = nullptr; uint32_t kind, count, n; void* bd; size_t nb;
    if (!get_ext(env, argv[0], &c) || !get_u32(env, argv[1], &kind) || !get_u32(env, argv[3], &count) || !get_u32(env, argv[4], &n) ||
        !get_bytes(env, argv[5], &bd, &nb) || nb < 48)
        return throw_type(env, "gridBuild(ctx, kind, primsBuf|null, count, nSlabs, Float64Array(6))");
    get_ext(env, argv[2], &pb);
    mirt_grid_build_desc d;
    memset(&d, 0, sizeof d);
    d.struct_size = sizeof d; d.kind = kind; d.count = count; d.n_slabs = n; d.prims_f64 = (mirt_buf*)pb;
    memcpy(d.bounds, bd, 48);
    mirt_buf *off = nullptr, *ord = nullptr; uint32_t total = 0;
    int rc = mirt_grid_build((mirt_ctx*)c, &d, &off, &ord, &total);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    napi_value o;
    napi_create_object(env, &o);
    napi_set_named_property(env, o, "offsets", mk_ext(env, off));
    napi_set_named_property(env, o, "order", mk_ext(env, ord));
    napi_set_named_property(env, o, "total", mk_num(env, total));
    return o;
}
// gridGatherTriangles(ctx, order, total, posF64Buf, norF64Buf|null, Int32Array ops, Float64Array vecs, padW) -> {pos, nor}
napi_value GridGatherTriangles(napi_env env, napi_callback_info info) {
    ARGS(8);
    void *c, *ord, *pb, *nb = nullptr; uint32_t total; void *ops, *vecs; size_t nops, nvecs; double pad;
    if (!get_ext(env, argv[0], &c) || !get_ext(env, argv[1], &ord) || !get_u32(env, argv[2], &total) || !get_ext(env, argv[3], &pb) ||
        !get_bytes(env, argv[5], &ops, &nops) || !get_bytes(env, argv[6], &vecs, &nvecs) || !get_f64(env, argv[7], &pad))
        return throw_type(env, "gridGatherTriangles(ctx, order, total, posBuf, norBuf|null, Int32Array, Float64Array, padW)");
    get_ext(env, argv[4], &nb);
    const uint32_t nsteps = (uint32_t)(nops / 4);
    if (nvecs < (size_t)nsteps * 24) return throw_type(env, "gridGatherTriangles: 3 doubles per step");
    mirt_buf *po = nullptr, *no = nullptr;
    int rc = mirt_grid_gather_triangles((mirt_ctx*)c, (mirt_buf*)ord, total, (mirt_buf*)pb, (mirt_buf*)nb, nsteps, (const int32_t*)ops, (const double*)vecs,
                                        (float)pad, &po, nb ? &no : nullptr);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    napi_value o;
    napi_create_object(env, &o);
    napi_set_named_property(env, o, "pos", mk_ext(env, po));
    if (no) napi_set_named_property(env, o, "nor", mk_ext(env, no));
    return o;
}
napi_value GridGather1(napi_env env, napi_callback_info info, bool spheres) {
    ARGS(4);
    void *c, *ord, *in; uint32_t total;
    if (!get_ext(env, argv[0], &c) || !get_ext(env, argv[1], &ord) || !get_u32(env, argv[2], &total) || !get_ext(env, argv[3], &in))
        return throw_type(env, "gridGather{Spheres,U32}(ctx, order, total, inBuf)");
    mirt_buf* out = nullptr;
    int rc = spheres ? mirt_grid_gather_spheres((mirt_ctx*)c, (mirt_buf*)ord, total, (mirt_buf*)in, &out)
                     : mirt_grid_gather_u32((mirt_ctx*)c, (mirt_buf*)ord, total, (mirt_buf*)in, &out);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return mk_ext(env, out);
}
napi_value GridGatherSpheres(napi_env env, napi_callback_info info) { return GridGather1(env, info, true); }
napi_value GridGatherU32(napi_env env, napi_callback_info info) { return GridGather1(env, info, false); }

napi_value TimerStart(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* c;
    if (!get_ext(env, argv[0], &c)) return throw_type(env, "timerStart(ctx)");
    int rc = mirt_timer_start((mirt_ctx*)c);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return undef(env);
}
napi_value TimerStopMs(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* c;
    if (!get_ext(env, argv[0], &c)) return throw_type(env, "timerStopMs(ctx)");
    float ms = 0;
    int rc = mirt_timer_stop_ms((mirt_ctx*)c, &ms);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return mk_num(env, ms);
}

// meshIngest(ctx, {nVertices, nCorners, firstCorner, model: Float32Array(16), normalMat: Float32Array(9), positions, normals, indices?}, posOut, norOut, bounds)
napi_value MeshIngest(napi_env env, napi_callback_info info) {
    ARGS(5);
    void *c, *po, *no, *bo;
    if (!get_ext(env, argv[0], &c) || !get_ext(env, argv[2], &po) || !get_ext(env, argv[3], &no) || !get_ext(env, argv[4], &bo))
        return throw_type(env, "meshIngest(ctx, desc, posOut, norOut, bounds)");
    mirt_mesh_ingest_desc d;
    memset(&d, 0, sizeof d);
    d.struct_size = sizeof d;
    if (!prop_u32(env, argv[1], "nVertices", &d.n_vertices) || !prop_u32(env, argv[1], "nCorners", &d.n_corners) || !prop_u32(env, argv[1], "firstCorner", &d.first_corner) ||
        !prop_floats(env, argv[1], "model", d.model, 16) || !prop_floats(env, argv[1], "normalMat", d.normal_mat, 9))
        return throw_type(env, "meshIngest: desc needs nVertices, nCorners, firstCorner, model (16 floats), normalMat (9 floats)");
    d.positions_f64 = prop_buf(env, argv[1], "positions");
    d.normals_f64 = prop_buf(env, argv[1], "normals");
    d.indices_u32 = prop_buf(env, argv[1], "indices");
    int rc = mirt_mesh_ingest((mirt_ctx*)c, &d, (mirt_buf*)po, (mirt_buf*)no, (mirt_buf*)bo);
    if (rc) return throw_mirt(env, rc, (mirt_ctx*)c);
    return undef(env);
}

// ---- device groups: mirt_group_create / _ctx / _destroy / _finish, mirt_tile_rows, mirt_gather -----------------------------
napi_value GroupCreate(napi_env env, napi_callback_info info) {
    ARGS(1);
    uint32_t n = 0;
    if (napi_get_array_length(env, argv[0], &n) != napi_ok || n == 0) return throw_type(env, "groupCreate([device indices])");
    std::vector<int> ids(n);
    for (uint32_t i = 0; i < n; ++i) {
        napi_value e; uint32_t d;
        if (napi_get_element(env, argv[0], i, &e) != napi_ok || !get_u32(env, e, &d)) return throw_type(env, "groupCreate: device indices are integers");
        ids[i] = (int)d;
    }
    mirt_group* g = nullptr;
    int rc = mirt_group_create(ids.data(), (int)n, &g);
    if (rc) return throw_mirt(env, rc, nullptr);
    return mk_ext(env, g);
}
napi_value GroupCtx(napi_env env, napi_callback_info info) {
    ARGS(2);
    void* g; uint32_t i;
    if (!get_ext(env, argv[0], &g) || !get_u32(env, argv[1], &i)) return throw_type(env, "groupCtx(group, index)");
    mirt_ctx* c = mirt_group_ctx((mirt_group*)g, (int)i);
    if (!c) return throw_mirt(env, MIRT_E_ARG, nullptr);
    return mk_ext(env, c);
}
napi_value GroupDestroy(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* g;
    if (!get_ext(env, argv[0], &g)) return throw_type(env, "groupDestroy(group)");
    int rc = mirt_group_destroy((mirt_group*)g);
    if (rc) return throw_mirt(env, rc, nullptr);
    return undef(env);
}
napi_value GroupFinish(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* g;
    if (!get_ext(env, argv[0], &g)) return throw_type(env, "groupFinish(group)");
    int rc = mirt_group_finish((mirt_group*)g);
    if (rc) return throw_mirt(env, rc, nullptr);
    return undef(env);
}
napi_value TileRows(napi_env env, napi_callback_info info) {
    ARGS(3);
    uint32_t h, n, i, r0 = 0, nr = 0;
    if (!get_u32(env, argv[0], &h) || !get_u32(env, argv[1], &n) || !get_u32(env, argv[2], &i)) return throw_type(env, "tileRows(height, nTiles, index)");
    mirt_tile_rows(h, n, i, &r0, &nr);
    napi_value o;
    napi_create_object(env, &o);
    napi_set_named_property(env, o, "row0", mk_num(env, r0));
    napi_set_named_property(env, o, "nrows", mk_num(env, nr));
    return o;
}
napi_value Gather(napi_env env, napi_callback_info info) {
    ARGS(6);
    void* g; void* out; uint32_t n = 0, root = 0, use_rccl = 0;
    if (!get_ext(env, argv[0], &g) || napi_get_array_length(env, argv[1], &n) != napi_ok || !get_ext(env, argv[3], &out) ||
        !get_u32(env, argv[4], &root) || !get_u32(env, argv[5], &use_rccl))
        return throw_type(env, "gather(group, [tile buffers], [tile bytes], out, root, transport)");
    const int gsize = mirt_group_size((mirt_group*)g);
    if (gsize < 0) return throw_mirt(env, gsize, nullptr);
    if ((int)n != gsize) return throw_type(env, "gather: one tile buffer per context of the group");
    std::vector<mirt_buf*> tiles(n);
    std::vector<size_t> bytes(n);
    for (uint32_t i = 0; i < n; ++i) {
        napi_value e; void* b; double nb;
        if (napi_get_element(env, argv[1], i, &e) != napi_ok || !get_ext(env, e, &b)) return throw_type(env, "gather: tiles are buffers");
        tiles[i] = (mirt_buf*)b;
        if (napi_get_element(env, argv[2], i, &e) != napi_ok || !get_f64(env, e, &nb) || nb < 0) return throw_type(env, "gather: tile byte counts are numbers");
        bytes[i] = (size_t)nb;
    }
    int rc = mirt_gather((mirt_group*)g, tiles.data(), bytes.data(), (int)n, (mirt_buf*)out, (int)root, (int)use_rccl);
    if (rc) return throw_mirt(env, rc, mirt_group_ctx((mirt_group*)g, 0));
    return undef(env);
}

// mirt_abi_version / mirt_group_peer_access / mirt_gather_route: what a first run on N real devices is diagnosed from
napi_value AbiVersion(napi_env env, napi_callback_info) { return mk_num(env, mirt_abi_version()); }
napi_value GroupPeerAccess(napi_env env, napi_callback_info info) {
    ARGS(3);
    void* g; uint32_t i, j;
    if (!get_ext(env, argv[0], &g) || !get_u32(env, argv[1], &i) || !get_u32(env, argv[2], &j)) return throw_type(env, "groupPeerAccess(group, i, j)");
    int rc = mirt_group_peer_access((const mirt_group*)g, (int)i, (int)j);
    if (rc < 0) return throw_mirt(env, rc, nullptr);
    return mk_num(env, rc);
}
napi_value GatherRoute(napi_env env, napi_callback_info info) {
    ARGS(2);
    void* g; uint32_t t;
    if (!get_ext(env, argv[0], &g) || !get_u32(env, argv[1], &t)) return throw_type(env, "gatherRoute(group, tile)");
    int rc = mirt_gather_route((const mirt_group*)g, (int)t);
    if (rc < 0) return throw_mirt(env, rc, nullptr);
    return mk_num(env, rc);
}

napi_value Init(napi_env env, napi_value exports) {
    struct { const char* name; napi_callback fn; } fns[] = {
        {"deviceCount", DeviceCount}, {"deviceName", DeviceName}, {"version", Version},
        {"ctxCreate", CtxCreate}, {"ctxDestroy", CtxDestroy}, {"finish", Finish}, {"ctxSetFusion", CtxSetFusion}, {"ctxFusedPasses", CtxFusedPasses},
        {"bufCreate", BufCreate}, {"bufRelease", BufRelease}, {"bufSize", BufSize}, {"bufWrite", BufWrite}, {"bufRead", BufRead},
        {"programCheck", ProgramCheck}, {"programDialect", ProgramDialect}, {"kernelGet", KernelGet}, {"kernelRelease", KernelRelease}, {"kernelNumArgs", KernelNumArgs},
        {"kernelPreferredMultiple", KernelPreferredMultiple}, {"kernelSetArg", KernelSetArg}, {"enqueue", Enqueue},
        {"renderPass", RenderPass}, {"gridBuild", GridBuild}, {"gridGatherTriangles", GridGatherTriangles},
        {"gridGatherSpheres", GridGatherSpheres}, {"gridGatherU32", GridGatherU32}, {"seedFill", SeedFill}, {"zero", Zero}, {"timerStart", TimerStart}, {"timerStopMs", TimerStopMs},
        {"captureBegin", CaptureBegin}, {"captureEnd", CaptureEnd}, {"graphLaunch", GraphLaunch}, {"graphRelease", GraphRelease},
        {"groupCreate", GroupCreate}, {"groupCtx", GroupCtx}, {"groupDestroy", GroupDestroy}, {"groupFinish", GroupFinish}, {"tileRows", TileRows}, {"gather", Gather}, {"meshIngest", MeshIngest},
        {"abiVersion", AbiVersion}, {"groupPeerAccess", GroupPeerAccess}, {"gatherRoute", GatherRoute},
    };
    for (auto& f : fns) {
        napi_value fn;
        napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &fn);
        napi_set_named_property(env, exports, f.name, fn);
    }
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
