/*
 * jslp_napi.c -- thin synchronous N-API addon over the C ABI of include/jslp_engine.h.
 *
 * This is the binding the reference's TypeScript host calls (north star: "the TypeScript host keeps model
 * parsing, validation and the branch-and-bound tree on the CPU, calling through a thin N-API C-ABI addon").
 * Raw C N-API only (node_api.h; no node-addon-api, no node-gyp):
 *     gcc -O2 -shared -fPIC -I/usr/include/node addon/jslp_napi.c -o addon/jslp_napi.node -ldl
 * The engine library is dlopen'ed by path (`load(path)`), so the same addon binds the product
 * (jslpsolver_amd/csrc/libjslp_hip.so) and -- in tests only -- the CPU oracle; there is no fallback: every call
 * before a successful load() throws.  TypedArrays are passed zero-copy (napi_get_typedarray_info).
 */
#define _POSIX_C_SOURCE 200809L  /* clock_gettime under -std=c11 */
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/jslp_engine.h"

static struct {
    void* dl;
    const char* (*backend_name)(void);
    const char* (*last_error)(void);
    int (*device_count)(void);
    int (*create)(jslp_engine**, int, int32_t, int32_t, int32_t, double);
    void (*destroy)(jslp_engine*);
    int (*upload)(jslp_engine*, const double*, const int32_t*, const int32_t*, const int32_t*, int32_t);
    int (*set_optional)(jslp_engine*, int32_t, const double*);
    int (*get_optional)(jslp_engine*, double*, int32_t*);
    int (*simplex)(jslp_engine*, int, jslp_simplex_result*);
    int (*pivot)(jslp_engine*, int32_t, int32_t);
    int (*save)(jslp_engine*);
    int (*restore)(jslp_engine*);
    int (*add_cuts)(jslp_engine*, int32_t, const int8_t*, const int32_t*, const double*);
    int (*relax)(jslp_engine*, int32_t, const int8_t*, const int32_t*, const double*, int, jslp_simplex_result*, double*, int32_t*);
    int (*relax_batch)(jslp_engine*, int32_t, const int32_t*, const int8_t*, const int32_t*, const double*, int,
                       jslp_simplex_result*, double*, int32_t*, int32_t);
    void (*release_pooled)(void);
    int (*set_integer_variables)(jslp_engine*, const int32_t*, int32_t);
    int (*apply_mir_cuts)(jslp_engine*, int32_t*);
    int (*checkpoint_create)(jslp_engine*, int32_t*);
    int (*checkpoint_restore)(jslp_engine*, int32_t);
    int (*checkpoint_release)(jslp_engine*, int32_t);
    int (*relax_from)(jslp_engine*, int32_t, int32_t, const int32_t*, const int8_t*, const int32_t*, const double*, int,
                      jslp_simplex_result*, double*, int32_t*, int32_t);
    int (*dims)(const jslp_engine*, int32_t*, int32_t*, int32_t*);
    int (*read_rhs)(jslp_engine*, double*, int32_t*);
    int (*download)(jslp_engine*, double*, int32_t*, int32_t*, int32_t*, int32_t*);
    int (*pivot_trace)(jslp_engine*, int32_t*, int64_t, int64_t*);
    int (*host_matrix)(jslp_engine*, double**, int64_t*);
    int (*set_watched)(jslp_engine*, const int32_t*, int32_t);
    int (*relax_watched)(jslp_engine*, int32_t, const int8_t*, const int32_t*, const double*, int, jslp_simplex_result*, int32_t*, double*);
    int (*relax_batch_watched)(jslp_engine*, int32_t, const int32_t*, const int8_t*, const int32_t*, const double*, int, jslp_simplex_result*,
                               int32_t*, double*);
    int (*set_counting)(jslp_engine*, int);
    int (*get_counters)(jslp_engine*, jslp_work_counters*);
    int (*pool_create)(jslp_pool**, jslp_engine*, const int32_t*, int32_t);
    void (*pool_destroy)(jslp_pool*);
    int (*pool_size)(const jslp_pool*);
    int (*pool_sync_root)(jslp_pool*);
    int (*pool_relax_batch)(jslp_pool*, int32_t, const int32_t*, const int8_t*, const int32_t*, const double*, int,
                            jslp_simplex_result*, double*, int32_t*, int32_t);
    int (*pool_set_watched)(jslp_pool*, const int32_t*, int32_t);
    int (*pool_relax_batch_watched)(jslp_pool*, int32_t, const int32_t*, const int8_t*, const int32_t*, const double*, int,
                                    jslp_simplex_result*, int32_t*, double*);
    int32_t (*watched_count)(const jslp_engine*);
    int32_t (*pool_watched_count)(const jslp_pool*);
    int (*set_timing)(jslp_engine*, int);
    int (*get_timing)(jslp_engine*, double*, int64_t*, double*);
} L;

/* what a JS engine handle points at: the engine plus the dimensions it was created with (argument checks without a
   device round trip) */
typedef struct { jslp_engine* e; int32_t h0, w, cap, n_watched; } ebox;

#define THROW(env, msg)                       \
    do {                                      \
        napi_throw_error(env, "JSLP", msg);   \
        return NULL;                          \
    } while (0)
#define NAPI_OK(env, call)                                     \
    do {                                                       \
        if ((call) != napi_ok) THROW(env, "N-API call failed: " #call); \
    } while (0)
#define ENGINE_OK(env, rc, what)                                                     \
    do {                                                                             \
        const int _rc = (rc);                                                        \
        if (_rc != JSLP_OK) {                                                        \
            char _m[640];                                                            \
            snprintf(_m, sizeof _m, "%s failed (%d): %s", what, _rc, L.last_error()); \
            THROW(env, _m);                                                          \
        }                                                                            \
    } while (0)

static int get_args(napi_env env, napi_callback_info info, size_t n, napi_value* argv) {
    size_t argc = n;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < n) {
        napi_throw_type_error(env, "JSLP", "wrong number of arguments");
        return 0;
    }
    return 1;
}

/* typed array -> pointer + length; `null`/`undefined` -> NULL */
static int typed(napi_env env, napi_value v, napi_typedarray_type want, void** data, size_t* len) {
    napi_valuetype t;
    *data = NULL;
    *len = 0;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_null || t == napi_undefined) return 1;
    bool is;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) {
        napi_throw_type_error(env, "JSLP", "typed array expected");
        return 0;
    }
    napi_typedarray_type ty;
    napi_value ab;
    size_t off;
    if (napi_get_typedarray_info(env, v, &ty, len, data, &ab, &off) != napi_ok) return 0;
    if (ty != want) {
        napi_throw_type_error(env, "JSLP", "typed array of the wrong element type");
        return 0;
    }
    return 1;
}

static jslp_engine* handle(napi_env env, napi_value v) {
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, "JSLP", "engine handle expected");
        return NULL;
    }
    jslp_engine* e = ((ebox*)p)->e;
    if (!e) napi_throw_error(env, "JSLP", "engine already destroyed");
    return e;
}
static const ebox* box_of(napi_env env, napi_value v) {
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) return NULL;
    return (const ebox*)p;
}

static void finalize_handle(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    ebox* box = (ebox*)data;
    if (box->e && L.destroy) L.destroy(box->e);
    free(box);
}

static napi_value result_object(napi_env env, const jslp_simplex_result* r) {
    napi_value o, v;
    NAPI_OK(env, napi_create_object(env, &o));
#define SET_B(name, x) NAPI_OK(env, napi_get_boolean(env, (x) != 0, &v)); NAPI_OK(env, napi_set_named_property(env, o, name, v))
#define SET_I(name, x) NAPI_OK(env, napi_create_int32(env, (x), &v)); NAPI_OK(env, napi_set_named_property(env, o, name, v))
#define SET_D(name, x) NAPI_OK(env, napi_create_double(env, (x), &v)); NAPI_OK(env, napi_set_named_property(env, o, name, v))
    SET_B("feasible", r->feasible);
    SET_B("bounded", r->bounded);
    SET_B("optimal", r->optimal);
    SET_I("unboundedVarIndex", r->unbounded_var_index);
    SET_I("pivotsPhase1", r->pivots_phase1);
    SET_I("pivotsPhase2", r->pivots_phase2);
    SET_I("cyclePhase", r->cycle_phase);
    SET_I("cycleStart", r->cycle_start);
    SET_I("cycleLength", r->cycle_length);
    SET_I("height", r->height);
    SET_D("objCell", r->obj_cell);
    SET_D("evaluation", r->evaluation);
    return o;
}

/* Batch results, PACKED (round 5): building one JS object per node through N-API costs 12 napi_set_named_property calls each -- tens of
   microseconds for a 16-node batch, comparable to the batch's kernel --, so the batch entry points take two optional trailing typed
   arrays and fill them instead: Int32Array[n * 10] = feasible, bounded, optimal, unboundedVarIndex, pivotsPhase1, pivotsPhase2,
   cyclePhase, cycleStart, cycleLength, height and Float64Array[n * 2] = objCell, evaluation per node -- result_object's fields in
   result_object's order (host/gpu-tableau.js `unpackResults` makes the same objects from them in JavaScript).  Without the two arrays:
   the array of objects, as before. */
#define JSLP_RES_I32 10
#define JSLP_RES_F64 2
/*  the optional packed-result arrays are checked BEFORE the engine call: a bad argument throws before the work is done, not after it */
static bool packed_args_ok(napi_env env, int32_t n_nodes, napi_value packed_i, napi_value packed_f, const char* who) {
    void *pi = NULL, *pf = NULL;
    size_t ni = 0, nf = 0;
    if (!typed(env, packed_i, napi_int32_array, &pi, &ni) || !typed(env, packed_f, napi_float64_array, &pf, &nf)) return false;
    if ((pi || pf) && (!pi || !pf || ni < (size_t)n_nodes * JSLP_RES_I32 || nf < (size_t)n_nodes * JSLP_RES_F64)) {
        napi_throw_error(env, "JSLP", who);
        return false;
    }
    return true;
}
static napi_value batch_results(napi_env env, jslp_simplex_result* res, int32_t n_nodes, napi_value packed_i, napi_value packed_f, const char* who) {
    void *pi = NULL, *pf = NULL;
    size_t ni = 0, nf = 0;
    if (!typed(env, packed_i, napi_int32_array, &pi, &ni) || !typed(env, packed_f, napi_float64_array, &pf, &nf)) { free(res); return NULL; }
    if (pi || pf) {
        if (!pi || !pf || ni < (size_t)n_nodes * JSLP_RES_I32 || nf < (size_t)n_nodes * JSLP_RES_F64) {
            free(res);
            napi_throw_error(env, "JSLP", who);
            return NULL;
        }
        int32_t* I = (int32_t*)pi;
        double* F = (double*)pf;
        for (int32_t i = 0; i < n_nodes; i++, I += JSLP_RES_I32, F += JSLP_RES_F64) {
            const jslp_simplex_result* r = &res[i];
            I[0] = r->feasible != 0; I[1] = r->bounded != 0; I[2] = r->optimal != 0; I[3] = r->unbounded_var_index;
            I[4] = r->pivots_phase1; I[5] = r->pivots_phase2; I[6] = r->cycle_phase; I[7] = r->cycle_start; I[8] = r->cycle_length;
            I[9] = r->height;
            F[0] = r->obj_cell; F[1] = r->evaluation;
        }
        free(res);
        napi_value u;
        NAPI_OK(env, napi_get_undefined(env, &u));
        return u;
    }
    napi_value arr;
    if (napi_create_array_with_length(env, (size_t)n_nodes, &arr) != napi_ok) { free(res); THROW(env, "array"); }
    for (int32_t i = 0; i < n_nodes; i++) {
        napi_value ro = result_object(env, &res[i]);
        if (!ro) { free(res); return NULL; }
        napi_set_element(env, arr, (uint32_t)i, ro);
    }
    free(res);
    return arr;
}
/* the first n_min arguments are required, the rest (up to n_max) arrive as undefined when absent */
static int get_args_opt(napi_env env, napi_callback_info info, size_t n_min, size_t n_max, napi_value* argv) {
    size_t argc = n_max;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < n_min) {
        napi_throw_type_error(env, "JSLP", "wrong number of arguments");
        return 0;
    }
    return 1;
}

/* load(path) -> backend name */
static napi_value fn_load(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    char path[4096];
    size_t n;
    NAPI_OK(env, napi_get_value_string_utf8(env, argv[0], path, sizeof path, &n));
    void* dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!dl) {
        char m[4600];
        snprintf(m, sizeof m, "cannot load engine library %s: %s (there is no CPU fallback)", path, dlerror());
        THROW(env, m);
    }
    memset(&L, 0, sizeof L);
    L.dl = dl;
#define SYM(field, name)                                                  \
    do {                                                                  \
        *(void**)(&L.field) = dlsym(dl, name);                            \
        if (!L.field) { THROW(env, "engine library lacks symbol " name); } \
    } while (0)
    SYM(backend_name, "jslp_backend_name"); SYM(last_error, "jslp_last_error"); SYM(device_count, "jslp_device_count");
    SYM(create, "jslp_engine_create"); SYM(destroy, "jslp_engine_destroy"); SYM(upload, "jslp_engine_upload");
    SYM(set_optional, "jslp_engine_set_optional_objectives"); SYM(get_optional, "jslp_engine_get_optional_objectives");
    SYM(simplex, "jslp_engine_simplex"); SYM(pivot, "jslp_engine_pivot"); SYM(save, "jslp_engine_save");
    SYM(restore, "jslp_engine_restore"); SYM(add_cuts, "jslp_engine_add_cuts"); SYM(relax, "jslp_engine_relax");
    SYM(relax_batch, "jslp_engine_relax_batch"); SYM(dims, "jslp_engine_dims"); SYM(read_rhs, "jslp_engine_read_rhs");
    SYM(download, "jslp_engine_download"); SYM(pivot_trace, "jslp_engine_pivot_trace");
    SYM(release_pooled, "jslp_release_pooled_resources");
    SYM(set_integer_variables, "jslp_engine_set_integer_variables"); SYM(apply_mir_cuts, "jslp_engine_apply_mir_cuts");
    SYM(checkpoint_create, "jslp_engine_checkpoint_create"); SYM(checkpoint_restore, "jslp_engine_checkpoint_restore");
    SYM(checkpoint_release, "jslp_engine_checkpoint_release"); SYM(relax_from, "jslp_engine_relax_from");
    SYM(host_matrix, "jslp_engine_host_matrix"); SYM(set_watched, "jslp_engine_set_watched_variables");
    SYM(relax_watched, "jslp_engine_relax_watched"); SYM(relax_batch_watched, "jslp_engine_relax_batch_watched"); SYM(set_counting, "jslp_engine_set_counting");
    SYM(get_counters, "jslp_engine_get_counters"); SYM(pool_create, "jslp_pool_create"); SYM(pool_destroy, "jslp_pool_destroy");
    SYM(pool_size, "jslp_pool_size"); SYM(pool_sync_root, "jslp_pool_sync_root"); SYM(pool_relax_batch, "jslp_pool_relax_batch");
    SYM(pool_set_watched, "jslp_pool_set_watched_variables"); SYM(pool_relax_batch_watched, "jslp_pool_relax_batch_watched");
    SYM(watched_count, "jslp_engine_watched_count"); SYM(pool_watched_count, "jslp_pool_watched_count");
    SYM(set_timing, "jslp_engine_set_timing"); SYM(get_timing, "jslp_engine_get_timing");
    napi_value s;
    NAPI_OK(env, napi_create_string_utf8(env, L.backend_name(), NAPI_AUTO_LENGTH, &s));
    return s;
}

#define NEED_LIB(env) do { if (!L.dl) THROW(env, "engine library not loaded: call load(path) first"); } while (0)

static napi_value fn_device_count(napi_env env, napi_callback_info info) {
    (void)info;
    NEED_LIB(env);
    napi_value v;
    NAPI_OK(env, napi_create_int32(env, L.device_count(), &v));
    return v;
}

/* create(height, width, rowCapacity, precision, device) -> handle */
static napi_value fn_create(napi_env env, napi_callback_info info) {
    NEED_LIB(env);
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return NULL;
    int32_t h, w, cap, dev;
    double precision;
    NAPI_OK(env, napi_get_value_int32(env, argv[0], &h));
    NAPI_OK(env, napi_get_value_int32(env, argv[1], &w));
    NAPI_OK(env, napi_get_value_int32(env, argv[2], &cap));
    NAPI_OK(env, napi_get_value_double(env, argv[3], &precision));
    NAPI_OK(env, napi_get_value_int32(env, argv[4], &dev));
    ebox* box = (ebox*)calloc(1, sizeof *box);
    int rc = L.create(&box->e, dev, h, w, cap, precision);
    box->h0 = h; box->w = w; box->cap = cap;
    if (rc != JSLP_OK) free(box);
    ENGINE_OK(env, rc, "jslp_engine_create");
    napi_value ext;
    NAPI_OK(env, napi_create_external(env, box, finalize_handle, NULL, &ext));
    return ext;
}

static napi_value fn_destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    void* p = NULL;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
        ebox* box = (ebox*)p;
        if (box->e) { L.destroy(box->e); box->e = NULL; }
    }
    return NULL;
}

/* upload(h, Float64Array matrix, Int32Array varIndexByRow, Int32Array varIndexByCol, Int32Array unrestricted) */
static napi_value fn_upload(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void *m, *r, *c, *u;
    size_t nm, nr, nc, nu;
    if (!typed(env, argv[1], napi_float64_array, &m, &nm) || !typed(env, argv[2], napi_int32_array, &r, &nr) ||
        !typed(env, argv[3], napi_int32_array, &c, &nc) || !typed(env, argv[4], napi_int32_array, &u, &nu))
        return NULL;
    const ebox* b = box_of(env, argv[0]);
    if (!m || !r || !c) THROW(env, "upload: matrix, varIndexByRow and varIndexByCol are required");
    if (nm < (size_t)b->h0 * (size_t)b->w || nr < (size_t)b->h0 || nc < (size_t)b->w)
        THROW(env, "upload: arrays shorter than the tableau (height x width cells, height rows, width columns)");
    ENGINE_OK(env, L.upload(e, (const double*)m, (const int32_t*)r, (const int32_t*)c, (const int32_t*)u, (int32_t)nu),
              "jslp_engine_upload");
    return NULL;
}

/* setOptionalObjectives(h, n, Float64Array rows) -- rows holds n * width doubles */
static napi_value fn_set_optional(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    int32_t n, W;
    NAPI_OK(env, napi_get_value_int32(env, argv[1], &n));
    void* rows;
    size_t len;
    if (!typed(env, argv[2], napi_float64_array, &rows, &len)) return NULL;
    ENGINE_OK(env, L.dims(e, NULL, &W, NULL), "jslp_engine_dims");
    if (n < 0 || len < (size_t)n * (size_t)W) THROW(env, "setOptionalObjectives: rows shorter than n * width");
    ENGINE_OK(env, L.set_optional(e, n, (const double*)rows), "jslp_engine_set_optional_objectives");
    return NULL;
}

/* getOptionalObjectives(h, Float64Array rows) -> n */
static napi_value fn_get_optional(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void* rows;
    size_t len;
    if (!typed(env, argv[1], napi_float64_array, &rows, &len)) return NULL;
    int32_t n = 0, W;
    ENGINE_OK(env, L.dims(e, NULL, &W, NULL), "jslp_engine_dims");
    ENGINE_OK(env, L.get_optional(e, NULL, &n), "jslp_engine_get_optional_objectives");
    if (len < (size_t)n * (size_t)W) THROW(env, "getOptionalObjectives: rows shorter than n * width");
    ENGINE_OK(env, L.get_optional(e, (double*)rows, &n), "jslp_engine_get_optional_objectives");
    napi_value v;
    NAPI_OK(env, napi_create_int32(env, n, &v));
    return v;
}

static napi_value fn_simplex(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    bool check;
    NAPI_OK(env, napi_get_value_bool(env, argv[1], &check));
    jslp_simplex_result r;
    ENGINE_OK(env, L.simplex(e, check ? 1 : 0, &r), "jslp_engine_simplex");
    return result_object(env, &r);
}

static napi_value fn_pivot(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    int32_t r, c;
    NAPI_OK(env, napi_get_value_int32(env, argv[1], &r));
    NAPI_OK(env, napi_get_value_int32(env, argv[2], &c));
    ENGINE_OK(env, L.pivot(e, r, c), "jslp_engine_pivot");
    return NULL;
}

static napi_value fn_save(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    ENGINE_OK(env, L.save(e), "jslp_engine_save");
    return NULL;
}

static napi_value fn_restore(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    ENGINE_OK(env, L.restore(e), "jslp_engine_restore");
    return NULL;
}

/* addCuts(h, Int8Array type, Int32Array varIndex, Float64Array value) */
static napi_value fn_add_cuts(napi_env env, napi_callback_info info) {
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void *t, *v, *x;
    size_t nt, nv, nx;
    if (!typed(env, argv[1], napi_int8_array, &t, &nt) || !typed(env, argv[2], napi_int32_array, &v, &nv) ||
        !typed(env, argv[3], napi_float64_array, &x, &nx))
        return NULL;
    if (nt != nv || nv != nx) THROW(env, "addCuts: array lengths differ");
    ENGINE_OK(env, L.add_cuts(e, (int32_t)nt, (const int8_t*)t, (const int32_t*)v, (const double*)x), "jslp_engine_add_cuts");
    return NULL;
}

/* relax(h, type, varIndex, value, checkCycles, Float64Array rhsOut|null, Int32Array rowsOut|null) -> result */
static napi_value fn_relax(napi_env env, napi_callback_info info) {
    napi_value argv[7];
    if (!get_args(env, info, 7, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void *t, *v, *x, *rhs, *rows;
    size_t nt, nv, nx, nrhs, nrows;
    bool check;
    if (!typed(env, argv[1], napi_int8_array, &t, &nt) || !typed(env, argv[2], napi_int32_array, &v, &nv) ||
        !typed(env, argv[3], napi_float64_array, &x, &nx))
        return NULL;
    NAPI_OK(env, napi_get_value_bool(env, argv[4], &check));
    if (!typed(env, argv[5], napi_float64_array, &rhs, &nrhs) || !typed(env, argv[6], napi_int32_array, &rows, &nrows)) return NULL;
    if (nt != nv || nv != nx) THROW(env, "relax: array lengths differ");
    {   /* the engine writes `height after the cuts` entries, at most the row capacity */
        const ebox* b = box_of(env, argv[0]);
        if ((rhs && nrhs < (size_t)b->cap) || (rows && nrows < (size_t)b->cap)) THROW(env, "relax: output arrays shorter than the row capacity");
    }
    /* the engine writes `height after the cuts` entries: bounded by the created row capacity, which the host passed
       to create() and sized its output arrays by */
    jslp_simplex_result r;
    ENGINE_OK(env, L.relax(e, (int32_t)nt, (const int8_t*)t, (const int32_t*)v, (const double*)x, check ? 1 : 0, &r,
                           (double*)rhs, (int32_t*)rows), "jslp_engine_relax");
    return result_object(env, &r);
}

/* releasePooledResources(): free the engine resources parked by destroyed engines (jslp_release_pooled_resources) */
static napi_value fn_release_pooled(napi_env env, napi_callback_info info) {
    (void)info;
    NEED_LIB(env);
    L.release_pooled();
    return NULL;
}

/* setIntegerVariables(h, Int32Array varIndexes): variable.isInteger for the MIR cuts (cutting-strategies.ts:82-85) */
static napi_value fn_set_integer_variables(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void* v;
    size_t nv;
    if (!typed(env, argv[1], napi_int32_array, &v, &nv)) return NULL;
    ENGINE_OK(env, L.set_integer_variables(e, (const int32_t*)v, (int32_t)nv), "jslp_engine_set_integer_variables");
    return NULL;
}

/* applyMirCuts(h) -> number of rows appended   (Tableau.applyMIRCuts, cutting-strategies.ts:199-212) */
static napi_value fn_apply_mir_cuts(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    int32_t n = 0;
    ENGINE_OK(env, L.apply_mir_cuts(e, &n), "jslp_engine_apply_mir_cuts");
    napi_value v;
    NAPI_OK(env, napi_create_int32(env, n, &v));
    return v;
}

/* checkpointCreate(h) -> id   (incremental-branch-and-cut.ts:55-70, the device part) */
static napi_value fn_checkpoint_create(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    int32_t id = -1;
    ENGINE_OK(env, L.checkpoint_create(e, &id), "jslp_engine_checkpoint_create");
    napi_value v;
    NAPI_OK(env, napi_create_int32(env, id, &v));
    return v;
}

/* checkpointRestore(h, id) / checkpointRelease(h, id) */
static napi_value checkpoint_op(napi_env env, napi_callback_info info, int release) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    int32_t id;
    NAPI_OK(env, napi_get_value_int32(env, argv[1], &id));
    if (release) ENGINE_OK(env, L.checkpoint_release(e, id), "jslp_engine_checkpoint_release");
    else ENGINE_OK(env, L.checkpoint_restore(e, id), "jslp_engine_checkpoint_restore");
    return NULL;
}
static napi_value fn_checkpoint_restore(napi_env env, napi_callback_info info) { return checkpoint_op(env, info, 0); }
static napi_value fn_checkpoint_release(napi_env env, napi_callback_info info) { return checkpoint_op(env, info, 1); }

/* relaxFrom(h, checkpointId, type, varIndex, value, checkCycles, rhsOut, rowsOut) -> result
   one child of a checkpointed parent: restoreCheckpoint + addCutConstraints + simplex (incremental-branch-and-cut.ts:248-253) */
static napi_value fn_relax_from(napi_env env, napi_callback_info info) {
    napi_value argv[8];
    if (!get_args(env, info, 8, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    int32_t id;
    NAPI_OK(env, napi_get_value_int32(env, argv[1], &id));
    void *t, *v, *x, *rhs, *rows;
    size_t nt, nv, nx, nrhs, nrows;
    bool check;
    if (!typed(env, argv[2], napi_int8_array, &t, &nt) || !typed(env, argv[3], napi_int32_array, &v, &nv) ||
        !typed(env, argv[4], napi_float64_array, &x, &nx))
        return NULL;
    NAPI_OK(env, napi_get_value_bool(env, argv[5], &check));
    if (!typed(env, argv[6], napi_float64_array, &rhs, &nrhs) || !typed(env, argv[7], napi_int32_array, &rows, &nrows)) return NULL;
    if (nt != nv || nv != nx) THROW(env, "relaxFrom: array lengths differ");
    {
        const ebox* b = box_of(env, argv[0]);
        if (!rhs || !rows || nrhs < (size_t)b->cap || nrows < (size_t)b->cap) THROW(env, "relaxFrom: output arrays shorter than the row capacity");
    }
    const int32_t offs[2] = {0, (int32_t)nt};
    /* one node: the stride only has to cover the row capacity, which is what the host sized its arrays by */
    const int32_t stride = (int32_t)(nrhs < nrows ? nrhs : nrows);
    jslp_simplex_result r;
    ENGINE_OK(env, L.relax_from(e, id, 1, offs, (const int8_t*)t, (const int32_t*)v, (const double*)x, check ? 1 : 0, &r,
                                (double*)rhs, (int32_t*)rows, stride), "jslp_engine_relax_from");
    return result_object(env, &r);
}

/* relaxBatch(h, Int32Array offsets, type, varIndex, value, checkCycles, rhsOut|null, rowsOut|null, stride) -> [result] */
static napi_value fn_relax_batch(napi_env env, napi_callback_info info) {
    napi_value argv[11];
    if (!get_args_opt(env, info, 9, 11, argv)) return NULL;  /* + the two optional packed-result arrays (batch_results) */
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void *o, *t, *v, *x, *rhs, *rows;
    size_t no, nt, nv, nx, nrhs, nrows;
    bool check;
    int32_t stride;
    if (!typed(env, argv[1], napi_int32_array, &o, &no) || !typed(env, argv[2], napi_int8_array, &t, &nt) ||
        !typed(env, argv[3], napi_int32_array, &v, &nv) || !typed(env, argv[4], napi_float64_array, &x, &nx))
        return NULL;
    NAPI_OK(env, napi_get_value_bool(env, argv[5], &check));
    if (!typed(env, argv[6], napi_float64_array, &rhs, &nrhs) || !typed(env, argv[7], napi_int32_array, &rows, &nrows)) return NULL;
    NAPI_OK(env, napi_get_value_int32(env, argv[8], &stride));
    if (no < 1) THROW(env, "relaxBatch: offsets must hold n_nodes + 1 entries");
    const int32_t n_nodes = (int32_t)no - 1;
    if (nt != nv || nv != nx || (size_t)((const int32_t*)o)[n_nodes] != nt) THROW(env, "relaxBatch: offsets[n_nodes] must equal the length of the cut arrays");
    if (stride < box_of(env, argv[0])->cap) THROW(env, "relaxBatch: stride below the row capacity");
    if ((rhs && nrhs < (size_t)n_nodes * (size_t)stride) || (rows && nrows < (size_t)n_nodes * (size_t)stride))
        THROW(env, "relaxBatch: output arrays shorter than n_nodes * stride");
    if (!packed_args_ok(env, n_nodes, argv[9], argv[10], "relax_batch: packed result arrays shorter than n_nodes * 10 / n_nodes * 2")) return NULL;
    jslp_simplex_result* res = (jslp_simplex_result*)calloc((size_t)(n_nodes > 0 ? n_nodes : 1), sizeof *res);
    int rc = L.relax_batch(e, n_nodes, (const int32_t*)o, (const int8_t*)t, (const int32_t*)v, (const double*)x,
                           check ? 1 : 0, res, (double*)rhs, (int32_t*)rows, stride);
    if (rc != JSLP_OK) free(res);
    ENGINE_OK(env, rc, "jslp_engine_relax_batch");
    return batch_results(env, res, n_nodes, argv[9], argv[10], "relax_batch: packed result arrays shorter than n_nodes * 10 / n_nodes * 2");
}

/* relaxBatchWatched(h, Int32Array offsets, type, varIndex, value, checkCycles, Int32Array rowsOut|null, Float64Array valuesOut|null)
   -> [result]: node i's watched variables at [i * nWatched, (i + 1) * nWatched) (setWatchedVariables first) */
static napi_value fn_relax_batch_watched(napi_env env, napi_callback_info info) {
    napi_value argv[10];
    if (!get_args_opt(env, info, 8, 10, argv)) return NULL;  /* + the two optional packed-result arrays (batch_results) */
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void *o, *t, *v, *x, *wr, *wv;
    size_t no, nt, nv, nx, nwr, nwv;
    bool check;
    if (!typed(env, argv[1], napi_int32_array, &o, &no) || !typed(env, argv[2], napi_int8_array, &t, &nt) ||
        !typed(env, argv[3], napi_int32_array, &v, &nv) || !typed(env, argv[4], napi_float64_array, &x, &nx))
        return NULL;
    NAPI_OK(env, napi_get_value_bool(env, argv[5], &check));
    if (!typed(env, argv[6], napi_int32_array, &wr, &nwr) || !typed(env, argv[7], napi_float64_array, &wv, &nwv)) return NULL;
    if (no < 1) THROW(env, "relaxBatchWatched: offsets must hold n_nodes + 1 entries");
    const int32_t n_nodes = (int32_t)no - 1;
    if (nt != nv || nv != nx || (size_t)((const int32_t*)o)[n_nodes] != nt) THROW(env, "relaxBatchWatched: offsets[n_nodes] must equal the length of the cut arrays");
    /* (sized by what the LIBRARY will write -- jslp_engine_watched_count -- not by this binding's shadow of the last list it
       registered itself: another user of the same engine may have changed it) */
    const int32_t nw_ = L.watched_count(e);
    if (nw_ <= 0) THROW(env, "relaxBatchWatched: setWatchedVariables first");
    const size_t n_watched = (size_t)nw_;
    if ((wr && nwr < (size_t)n_nodes * n_watched) || (wv && nwv < (size_t)n_nodes * n_watched))
        THROW(env, "relaxBatchWatched: output arrays shorter than n_nodes * nWatched");
    if (!packed_args_ok(env, n_nodes, argv[8], argv[9], "relax_batch_watched: packed result arrays shorter than n_nodes * 10 / n_nodes * 2")) return NULL;
    jslp_simplex_result* res = (jslp_simplex_result*)calloc((size_t)(n_nodes > 0 ? n_nodes : 1), sizeof *res);
    int rc = L.relax_batch_watched(e, n_nodes, (const int32_t*)o, (const int8_t*)t, (const int32_t*)v, (const double*)x,
                                   check ? 1 : 0, res, (int32_t*)wr, (double*)wv);
    if (rc != JSLP_OK) free(res);
    ENGINE_OK(env, rc, "jslp_engine_relax_batch_watched");
    return batch_results(env, res, n_nodes, argv[8], argv[9], "relax_batch_watched: packed result arrays shorter than n_nodes * 10 / n_nodes * 2");
}

/* dims(h) -> {height, width, nVarIndexes} */
static napi_value fn_dims(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    int32_t H, W, N;
    ENGINE_OK(env, L.dims(e, &H, &W, &N), "jslp_engine_dims");
    napi_value o, v;
    NAPI_OK(env, napi_create_object(env, &o));
    SET_I("height", H);
    SET_I("width", W);
    SET_I("nVarIndexes", N);
    return o;
}

/* readRhs(h, Float64Array rhs|null, Int32Array rows|null) */
static napi_value fn_read_rhs(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void *rhs, *rows;
    size_t n1, n2;
    if (!typed(env, argv[1], napi_float64_array, &rhs, &n1) || !typed(env, argv[2], napi_int32_array, &rows, &n2)) return NULL;
    const ebox* b = box_of(env, argv[0]);
    if ((rhs && n1 < (size_t)b->cap) || (rows && n2 < (size_t)b->cap)) {  /* shorter than the capacity: check the live height */
        int32_t H;
        ENGINE_OK(env, L.dims(e, &H, NULL, NULL), "jslp_engine_dims");
        if ((rhs && n1 < (size_t)H) || (rows && n2 < (size_t)H)) THROW(env, "readRhs: output arrays shorter than the height");
    }
    ENGINE_OK(env, L.read_rhs(e, (double*)rhs, (int32_t*)rows), "jslp_engine_read_rhs");
    return NULL;
}

/* download(h, Float64Array matrix|null, Int32Array vibr|null, Int32Array vibc|null, Int32Array rbv|null, Int32Array cbv|null) */
static napi_value fn_download(napi_env env, napi_callback_info info) {
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void* p[5];
    size_t n[5];
    if (!typed(env, argv[1], napi_float64_array, &p[0], &n[0])) return NULL;
    for (int i = 1; i < 5; i++)
        if (!typed(env, argv[1 + i], napi_int32_array, &p[i], &n[i])) return NULL;
    int32_t H, W, N;
    ENGINE_OK(env, L.dims(e, &H, &W, &N), "jslp_engine_dims");
    if ((p[0] && n[0] < (size_t)H * W) || (p[1] && n[1] < (size_t)H) || (p[2] && n[2] < (size_t)W) ||
        (p[3] && n[3] < (size_t)N) || (p[4] && n[4] < (size_t)N))
        THROW(env, "download: output arrays too short");
    ENGINE_OK(env, L.download(e, (double*)p[0], (int32_t*)p[1], (int32_t*)p[2], (int32_t*)p[3], (int32_t*)p[4]),
              "jslp_engine_download");
    return NULL;
}

/* pivotTrace(h) -> Int32Array [r0, c0, r1, c1, ...] */
static napi_value fn_pivot_trace(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    int64_t n = 0;
    ENGINE_OK(env, L.pivot_trace(e, NULL, 0, &n), "jslp_engine_pivot_trace");
    napi_value ab, ta;
    void* data;
    NAPI_OK(env, napi_create_arraybuffer(env, (size_t)n * 8 + 8, &data, &ab));
    ENGINE_OK(env, L.pivot_trace(e, (int32_t*)data, n, &n), "jslp_engine_pivot_trace");
    NAPI_OK(env, napi_create_typedarray(env, napi_int32_array, (size_t)n * 2, ab, 0, &ta));
    return ta;
}

/* hostMatrix(h) -> Float64Array (height x width) over the engine's PINNED build buffer (jslp_engine_host_matrix): the host
   builds the tableau straight into DMA-able memory and passes the very array to upload().  The engine owns
   the memory: the binding detaches the array (detach()) before it destroys the engine. */
static void release_keep(napi_env env, void* data, void* hint) {
    (void)data;
    if (hint) napi_delete_reference(env, (napi_ref)hint);
}
static napi_value fn_host_matrix(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    double* m = NULL;
    int64_t n = 0;
    ENGINE_OK(env, L.host_matrix(e, &m, &n), "jslp_engine_host_matrix");
    napi_value ab, ta;
    /* the view keeps the engine handle alive: a garbage-collected handle destroys the engine (finalize_handle), which must
       not happen while JS can still reach the engine's memory through this array */
    napi_ref keep = NULL;
    NAPI_OK(env, napi_create_reference(env, argv[0], 1, &keep));
    NAPI_OK(env, napi_create_external_arraybuffer(env, m, (size_t)n * sizeof(double), release_keep, keep, &ab));
    NAPI_OK(env, napi_create_typedarray(env, napi_float64_array, (size_t)n, ab, 0, &ta));
    return ta;
}

/* detach(typedArray): the array (a hostMatrix() view) becomes zero-length, its memory goes back to the engine */
static napi_value fn_detach(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    bool is = false;
    if (napi_is_typedarray(env, argv[0], &is) != napi_ok || !is) THROW(env, "detach: typed array expected");
    napi_typedarray_type ty; size_t len, off; void* data; napi_value ab;
    NAPI_OK(env, napi_get_typedarray_info(env, argv[0], &ty, &len, &data, &ab, &off));
    NAPI_OK(env, napi_detach_arraybuffer(env, ab));
    return NULL;
}

/* setWatchedVariables(h, Int32Array varIndexes) */
static napi_value fn_set_watched(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void* v; size_t n;
    if (!typed(env, argv[1], napi_int32_array, &v, &n)) return NULL;
    ENGINE_OK(env, L.set_watched(e, (const int32_t*)v, (int32_t)n), "jslp_engine_set_watched_variables");
    {   /* relaxBatchWatched sizes its output checks by this */
        void* p = NULL;
        if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) ((ebox*)p)->n_watched = (int32_t)n;
    }
    return NULL;
}

/* relaxWatched(h, type, varIndex, value, checkCycles, Int32Array watchedRow, Float64Array watchedValue) -> result */
static napi_value fn_relax_watched(napi_env env, napi_callback_info info) {
    napi_value argv[7];
    if (!get_args(env, info, 7, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void *t, *v, *x, *wr, *wv;
    size_t nt, nv, nx, nwr, nwv;
    bool check;
    if (!typed(env, argv[1], napi_int8_array, &t, &nt) || !typed(env, argv[2], napi_int32_array, &v, &nv) ||
        !typed(env, argv[3], napi_float64_array, &x, &nx))
        return NULL;
    NAPI_OK(env, napi_get_value_bool(env, argv[4], &check));
    if (!typed(env, argv[5], napi_int32_array, &wr, &nwr) || !typed(env, argv[6], napi_float64_array, &wv, &nwv)) return NULL;
    if (nt != nv || nv != nx) THROW(env, "relaxWatched: array lengths differ");
    /* the engine writes one entry per watched variable: the host sized both arrays by the list it registered, and the
       engine refuses lists longer than the row capacity */
    {
        const int32_t n_watched = L.watched_count(e);  /* (the library's own count: see relaxBatchWatched) */
        if (n_watched <= 0) THROW(env, "relaxWatched: setWatchedVariables first");
        if (!wr || !wv || nwr != nwv || nwr < (size_t)n_watched)
            THROW(env, "relaxWatched: watchedRow and watchedValue must have one entry per watched variable");
    }
    jslp_simplex_result r;
    ENGINE_OK(env, L.relax_watched(e, (int32_t)nt, (const int8_t*)t, (const int32_t*)v, (const double*)x, check ? 1 : 0, &r,
                                   (int32_t*)wr, (double*)wv), "jslp_engine_relax_watched");
    return result_object(env, &r);
}

/* setCounting(h, on) / getCounters(h) -> {relaxations, simplexCalls, pivots, gatedCells, gatedRows, restoredRows, cutRows, heightSum} */
static napi_value fn_set_counting(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    bool on;
    NAPI_OK(env, napi_get_value_bool(env, argv[1], &on));
    ENGINE_OK(env, L.set_counting(e, on ? 1 : 0), "jslp_engine_set_counting");
    return NULL;
}
static napi_value fn_get_counters(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    jslp_work_counters c;
    ENGINE_OK(env, L.get_counters(e, &c), "jslp_engine_get_counters");
    napi_value o, v;
    NAPI_OK(env, napi_create_object(env, &o));
    SET_D("relaxations", (double)c.relaxations); SET_D("simplexCalls", (double)c.simplex_calls); SET_D("pivots", (double)c.pivots);
    SET_D("gatedCells", (double)c.gated_cells); SET_D("gatedRows", (double)c.gated_rows); SET_D("restoredRows", (double)c.restored_rows);
    SET_D("cutRows", (double)c.cut_rows); SET_D("heightSum", (double)c.height_sum);
    SET_D("residentAborts", (double)c.resident_aborts); SET_D("residentHandovers", (double)c.resident_handovers);
    SET_D("residentLaunches", (double)c.resident_launches); SET_D("residentRefusals", (double)c.resident_refusals);
    SET_D("nodeQueueLaunches", (double)c.node_queue_launches);
    SET_D("residentFetchRetries", (double)c.resident_fetch_retries);
    return o;
}

/* ---- device pool (jslp_pool_*) ---- */
typedef struct { jslp_pool* p; int32_t cap; int32_t n_watched; napi_ref primary; } pbox;
/* The pool's members must go before the primary engine does, and finalisers run in no particular order: the box holds a
   reference to the primary's handle, so the primary outlives the pool whichever way the pool ends -- poolDestroy (what the
   binding calls before it destroys the primary) or this finaliser (a tableau that was simply dropped: the N - 1 member
   engines, their streams and their worker threads must not live on for the rest of the process). */
static void pool_box_close(napi_env env, pbox* box) {
    if (box->p) { L.pool_destroy(box->p); box->p = NULL; }
    if (box->primary) { napi_delete_reference(env, box->primary); box->primary = NULL; }
}
static void finalize_pool(napi_env env, void* data, void* hint) {
    (void)hint;
    pool_box_close(env, (pbox*)data);
    free(data);
}
static jslp_pool* pool_handle(napi_env env, napi_value v, int32_t* cap) {
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((pbox*)p)->p) {
        napi_throw_type_error(env, "JSLP", "pool handle expected (or pool already destroyed)");
        return NULL;
    }
    if (cap) *cap = ((pbox*)p)->cap;
    return ((pbox*)p)->p;
}
/* poolCreate(h, Int32Array devices) -> pool handle; devices[0] = the primary's device, ordinals may repeat */
static napi_value fn_pool_create(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    void* d; size_t n;
    if (!typed(env, argv[1], napi_int32_array, &d, &n)) return NULL;
    if (!d || n < 1) THROW(env, "poolCreate: at least one device ordinal");
    pbox* box = (pbox*)calloc(1, sizeof *box);
    box->cap = box_of(env, argv[0])->cap;
    int rc = L.pool_create(&box->p, e, (const int32_t*)d, (int32_t)n);
    if (rc != JSLP_OK) free(box);
    ENGINE_OK(env, rc, "jslp_pool_create");
    napi_value ext;
    if (napi_create_reference(env, argv[0], 1, &box->primary) != napi_ok) box->primary = NULL;
    if (napi_create_external(env, box, finalize_pool, NULL, &ext) != napi_ok) {
        pool_box_close(env, box);
        free(box);
        THROW(env, "poolCreate: napi_create_external failed");
    }
    return ext;
}
static napi_value fn_pool_destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    void* p = NULL;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
        pool_box_close(env, (pbox*)p);
    }
    return NULL;
}
static napi_value fn_pool_size(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    jslp_pool* p = pool_handle(env, argv[0], NULL);
    if (!p) return NULL;
    napi_value v;
    NAPI_OK(env, napi_create_int32(env, L.pool_size(p), &v));
    return v;
}
static napi_value fn_pool_sync_root(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    jslp_pool* p = pool_handle(env, argv[0], NULL);
    if (!p) return NULL;
    ENGINE_OK(env, L.pool_sync_root(p), "jslp_pool_sync_root");
    return NULL;
}
/* poolRelaxBatch(pool, Int32Array offsets, type, varIndex, value, checkCycles, rhsOut|null, rowsOut|null, stride) -> [result] */
static napi_value fn_pool_relax_batch(napi_env env, napi_callback_info info) {
    napi_value argv[11];
    if (!get_args_opt(env, info, 9, 11, argv)) return NULL;  /* + the two optional packed-result arrays (batch_results) */
    int32_t cap = 0;
    jslp_pool* p = pool_handle(env, argv[0], &cap);
    if (!p) return NULL;
    void *o, *t, *v, *x, *rhs, *rows;
    size_t no, nt, nv, nx, nrhs, nrows;
    bool check;
    int32_t stride;
    if (!typed(env, argv[1], napi_int32_array, &o, &no) || !typed(env, argv[2], napi_int8_array, &t, &nt) ||
        !typed(env, argv[3], napi_int32_array, &v, &nv) || !typed(env, argv[4], napi_float64_array, &x, &nx))
        return NULL;
    NAPI_OK(env, napi_get_value_bool(env, argv[5], &check));
    if (!typed(env, argv[6], napi_float64_array, &rhs, &nrhs) || !typed(env, argv[7], napi_int32_array, &rows, &nrows)) return NULL;
    NAPI_OK(env, napi_get_value_int32(env, argv[8], &stride));
    if (no < 1) THROW(env, "poolRelaxBatch: offsets must hold n_nodes + 1 entries");
    const int32_t n_nodes = (int32_t)no - 1;
    if (nt != nv || nv != nx || (size_t)((const int32_t*)o)[n_nodes] != nt) THROW(env, "poolRelaxBatch: offsets[n_nodes] must equal the length of the cut arrays");
    if (stride < cap) THROW(env, "poolRelaxBatch: stride below the row capacity");
    if ((rhs && nrhs < (size_t)n_nodes * (size_t)stride) || (rows && nrows < (size_t)n_nodes * (size_t)stride))
        THROW(env, "poolRelaxBatch: output arrays shorter than n_nodes * stride");
    if (!packed_args_ok(env, n_nodes, argv[9], argv[10], "pool_relax_batch: packed result arrays shorter than n_nodes * 10 / n_nodes * 2")) return NULL;
    jslp_simplex_result* res = (jslp_simplex_result*)calloc((size_t)(n_nodes > 0 ? n_nodes : 1), sizeof *res);
    int rc = L.pool_relax_batch(p, n_nodes, (const int32_t*)o, (const int8_t*)t, (const int32_t*)v, (const double*)x,
                                check ? 1 : 0, res, (double*)rhs, (int32_t*)rows, stride);
    if (rc != JSLP_OK) free(res);
    ENGINE_OK(env, rc, "jslp_pool_relax_batch");
    return batch_results(env, res, n_nodes, argv[9], argv[10], "pool_relax_batch: packed result arrays shorter than n_nodes * 10 / n_nodes * 2");
}

/* poolSetWatchedVariables(pool, Int32Array varIndexes): jslp_pool_set_watched_variables (every member) */
static napi_value fn_pool_set_watched(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    jslp_pool* p = pool_handle(env, argv[0], NULL);
    if (!p) return NULL;
    void* v; size_t n;
    if (!typed(env, argv[1], napi_int32_array, &v, &n)) return NULL;
    ENGINE_OK(env, L.pool_set_watched(p, (const int32_t*)v, (int32_t)n), "jslp_pool_set_watched_variables");
    {
        void* bp = NULL;
        if (napi_get_value_external(env, argv[0], &bp) == napi_ok && bp) ((pbox*)bp)->n_watched = (int32_t)n;
    }
    return NULL;
}

/* poolRelaxBatchWatched(pool, Int32Array offsets, type, varIndex, value, checkCycles, Int32Array rowsOut, Float64Array valuesOut)
   -> [result]: the compact read-back (mip-utils.ts:43-61, 100-126 need only the integer variables' cells) over every member */
static napi_value fn_pool_relax_batch_watched(napi_env env, napi_callback_info info) {
    napi_value argv[10];
    if (!get_args_opt(env, info, 8, 10, argv)) return NULL;  /* + the two optional packed-result arrays (batch_results) */
    jslp_pool* p = pool_handle(env, argv[0], NULL);
    if (!p) return NULL;
    void *o, *t, *v, *x, *wr, *wv, *bp = NULL;
    size_t no, nt, nv, nx, nwr, nwv;
    bool check;
    if (!typed(env, argv[1], napi_int32_array, &o, &no) || !typed(env, argv[2], napi_int8_array, &t, &nt) ||
        !typed(env, argv[3], napi_int32_array, &v, &nv) || !typed(env, argv[4], napi_float64_array, &x, &nx))
        return NULL;
    NAPI_OK(env, napi_get_value_bool(env, argv[5], &check));
    if (!typed(env, argv[6], napi_int32_array, &wr, &nwr) || !typed(env, argv[7], napi_float64_array, &wv, &nwv)) return NULL;
    if (no < 1) THROW(env, "poolRelaxBatchWatched: offsets must hold n_nodes + 1 entries");
    const int32_t n_nodes = (int32_t)no - 1;
    if (nt != nv || nv != nx || (size_t)((const int32_t*)o)[n_nodes] != nt) THROW(env, "poolRelaxBatchWatched: offsets[n_nodes] must equal the length of the cut arrays");
    NAPI_OK(env, napi_get_value_external(env, argv[0], &bp));
    (void)bp;
    const int32_t nwp_ = L.pool_watched_count(p);  /* every member's count, -1 when they differ (not the JS-side shadow) */
    if (nwp_ == 0) THROW(env, "poolRelaxBatchWatched: poolSetWatchedVariables first");
    if (nwp_ < 0) THROW(env, "poolRelaxBatchWatched: the members' watched variables differ (poolSetWatchedVariables sets them all)");
    const size_t n_watched = (size_t)nwp_;
    if ((wr && nwr < (size_t)n_nodes * n_watched) || (wv && nwv < (size_t)n_nodes * n_watched))
        THROW(env, "poolRelaxBatchWatched: output arrays shorter than n_nodes * nWatched");
    if (!packed_args_ok(env, n_nodes, argv[8], argv[9], "pool_relax_batch_watched: packed result arrays shorter than n_nodes * 10 / n_nodes * 2")) return NULL;
    jslp_simplex_result* res = (jslp_simplex_result*)calloc((size_t)(n_nodes > 0 ? n_nodes : 1), sizeof *res);
    int rc = L.pool_relax_batch_watched(p, n_nodes, (const int32_t*)o, (const int8_t*)t, (const int32_t*)v, (const double*)x,
                                        check ? 1 : 0, res, (int32_t*)wr, (double*)wv);
    if (rc != JSLP_OK) free(res);
    ENGINE_OK(env, rc, "jslp_pool_relax_batch_watched");
    return batch_results(env, res, n_nodes, argv[8], argv[9], "pool_relax_batch_watched: packed result arrays shorter than n_nodes * 10 / n_nodes * 2");
}

/* ---- where a Solve() spends its time inside the binding ------------------------------------------------------------------------
   Every exported function runs behind one trampoline that adds its wall time (CLOCK_MONOTONIC) and a call to its entry of `fns`;
   `timings(reset)` returns { name: [milliseconds, calls], ... } for the entries that were called.  Two clock reads per call (~40 ns):
   always on.  bench.py's `dropin_js` leg and tools/shim_profile.js build their per-phase tables (create + pin / upload / simplex /
   read-back / release) from it; the device time inside simplex() is jslp_engine_get_timing's (HIP events), exported as `deviceMs`. */
typedef struct { const char* name; napi_callback fn; double ns; long long calls; } fn_entry;
static napi_value fn_timings(napi_env env, napi_callback_info info);
static napi_value fn_device_ms(napi_env env, napi_callback_info info);
static napi_value timed_cb(napi_env env, napi_callback_info info) {
    size_t argc = 0;
    void* data = NULL;
    if (napi_get_cb_info(env, info, &argc, NULL, NULL, &data) != napi_ok || !data) THROW(env, "internal: function entry missing");
    fn_entry* fe = (fn_entry*)data;  /* (this env's own copy of the table: see init) */
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    napi_value r = fe->fn(env, info);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    fe->ns += (double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec);
    fe->calls += 1;
    return r;
}

static fn_entry fns[] = {
        {"load", fn_load}, {"deviceCount", fn_device_count}, {"create", fn_create}, {"destroy", fn_destroy},
        {"upload", fn_upload}, {"setOptionalObjectives", fn_set_optional}, {"getOptionalObjectives", fn_get_optional},
        {"simplex", fn_simplex}, {"pivot", fn_pivot}, {"save", fn_save}, {"restore", fn_restore},
        {"addCuts", fn_add_cuts}, {"relax", fn_relax}, {"relaxBatch", fn_relax_batch}, {"dims", fn_dims},
        {"readRhs", fn_read_rhs}, {"download", fn_download}, {"pivotTrace", fn_pivot_trace},
        {"releasePooledResources", fn_release_pooled}, {"setIntegerVariables", fn_set_integer_variables}, {"applyMirCuts", fn_apply_mir_cuts},
        {"checkpointCreate", fn_checkpoint_create}, {"checkpointRestore", fn_checkpoint_restore},
        {"checkpointRelease", fn_checkpoint_release}, {"relaxFrom", fn_relax_from},
        {"hostMatrix", fn_host_matrix}, {"detach", fn_detach}, {"setWatchedVariables", fn_set_watched}, {"relaxWatched", fn_relax_watched}, {"relaxBatchWatched", fn_relax_batch_watched},
        {"setCounting", fn_set_counting}, {"getCounters", fn_get_counters},
        {"poolCreate", fn_pool_create}, {"poolDestroy", fn_pool_destroy}, {"poolSize", fn_pool_size},
        {"poolSyncRoot", fn_pool_sync_root}, {"poolRelaxBatch", fn_pool_relax_batch},
        {"poolSetWatchedVariables", fn_pool_set_watched}, {"poolRelaxBatchWatched", fn_pool_relax_batch_watched},
        {"deviceMs", fn_device_ms},
};
#define N_FNS (sizeof fns / sizeof fns[0])

/* timings(reset = false) -> { name: [ms, calls] } of every binding function called since the last reset */
static napi_value fn_timings(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    bool reset = false;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) == napi_ok && argc >= 1) napi_get_value_bool(env, argv[0], &reset);
    fn_entry* tab = NULL;
    if (napi_get_instance_data(env, (void**)&tab) != napi_ok || !tab) THROW(env, "timings: no table for this environment");
    napi_value o;
    NAPI_OK(env, napi_create_object(env, &o));
    for (size_t i = 0; i < N_FNS; i++) {
        if (tab[i].calls == 0) continue;
        napi_value pair, ms, calls;
        NAPI_OK(env, napi_create_array_with_length(env, 2, &pair));
        NAPI_OK(env, napi_create_double(env, tab[i].ns / 1e6, &ms));
        NAPI_OK(env, napi_create_double(env, (double)tab[i].calls, &calls));
        napi_set_element(env, pair, 0, ms);
        napi_set_element(env, pair, 1, calls);
        NAPI_OK(env, napi_set_named_property(env, o, tab[i].name, pair));
        if (reset) { tab[i].ns = 0; tab[i].calls = 0; }
    }
    return o;
}

/* deviceMs(engine, enable?) -> milliseconds the engine's stream spent inside simplex() / relax() since timing was switched on
   (HIP events around the kernels: jslp_engine_get_timing); deviceMs(engine, true) switches it on and zeroes the sum */
static napi_value fn_device_ms(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < 1) THROW(env, "deviceMs(engine[, enable])");
    jslp_engine* e = handle(env, argv[0]);
    if (!e) return NULL;
    if (!L.set_timing || !L.get_timing) THROW(env, "deviceMs: the loaded library has no timing entry points");
    if (argc >= 2) {
        bool on = false;
        napi_get_value_bool(env, argv[1], &on);
        ENGINE_OK(env, L.set_timing(e, on ? 1 : 0), "set_timing");
    }
    double upd = 0, total = 0;
    int64_t launches = 0;
    ENGINE_OK(env, L.get_timing(e, &upd, &launches, &total), "get_timing");
    napi_value v;
    NAPI_OK(env, napi_create_double(env, total, &v));
    return v;
}

static void free_table(napi_env env, void* data, void* hint) { (void)env; (void)hint; free(data); }
static napi_value init(napi_env env, napi_value exports) {
    /*  the timing table is PER ENVIRONMENT (napi_set_instance_data): the addon can be loaded from several Node worker threads of
       one process, which used to race on one static table -- and timings(true) in one thread reset another's figures */
    fn_entry* tab = (fn_entry*)malloc(sizeof fns);
    if (!tab) return NULL;
    memcpy(tab, fns, sizeof fns);
    if (napi_set_instance_data(env, tab, free_table, NULL) != napi_ok) { free(tab); return NULL; }
    for (size_t i = 0; i < N_FNS; i++) {
        napi_value f;
        if (napi_create_function(env, tab[i].name, NAPI_AUTO_LENGTH, timed_cb, &tab[i], &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, tab[i].name, f) != napi_ok) return NULL;
    }
    napi_value f;  /* (not timed itself) */
    if (napi_create_function(env, "timings", NAPI_AUTO_LENGTH, fn_timings, NULL, &f) != napi_ok) return NULL;
    if (napi_set_named_property(env, exports, "timings", f) != napi_ok) return NULL;
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
