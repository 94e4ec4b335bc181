/* N-API addon: the hashing / constraint workers of the reference's browser SDK backed by libaero_stark.so.
 *
 * The SDK's workers are thin shims around two wasm entry points that take one bincode message and post one back
 * (aero-sdk/src/hashing_worker.ts, constraints_worker.ts -> miden-wasm hashing_entry_point / constraint_entry_point,
 * hashing_worker.rs:28-42, constraints_worker.rs:81-97). This addon gives a Node host the same two functions over the
 * library's message-level entry points (include/aero_stark.h: aero_worker_hash_rows, aero_worker_eval_constraints) plus the
 * ProverOutput message (aero_prover_output). The library is bound with dlopen, so the addon builds without it:
 *
 *   gcc -shared -fPIC -O2 -I/usr/include/node bindings/node/aero_worker.c -o bindings/node/aero_worker.node -ldl
 *
 *   const aero = require('./aero_worker.node');
 *   const h = aero.open('/path/to/libaero_stark.so', 0);          // device id; -1 = no GPU context (proverOutput only)
 *   const result = aero.hashRows(h, payload);                      // Uint8Array bincode(HashingWorkItem) -> Buffer bincode(HashingResult)
 *   const cols = aero.evalConstraints(h, payload, [auxWidth, auxRands, auxDegree] or null);
 *   const out = aero.proverOutput(h, proofBytes, inputBytes);      // bincode(ProverOutput)
 *   aero.close(h);
 * A handle belongs to the thread that opened it (one context per worker thread, like one wasm instance per web worker). */
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct aero_ctx aero_ctx;
typedef struct { uint32_t aux_width, aux_rands, aux_degree; } aero_fib_air;

typedef struct {
    void* lib;
    aero_ctx* ctx;
    int32_t (*ctx_create)(int32_t, aero_ctx**);
    void (*ctx_destroy)(aero_ctx*);
    const char* (*last_error)(const aero_ctx*);
    void (*free_buf)(void*);
    int32_t (*hash_rows)(aero_ctx*, const uint8_t*, size_t, uint8_t**, size_t*);
    int32_t (*eval_constraints)(aero_ctx*, const uint8_t*, size_t, const aero_fib_air*, uint8_t**, size_t*);
    int32_t (*prover_output)(const uint8_t*, size_t, const uint8_t*, size_t, uint8_t**, size_t*, char*, size_t);
} handle_t;

#define CHECK(call) do { if ((call) != napi_ok) { napi_throw_error(env, NULL, "aero_worker: N-API call failed: " #call); return NULL; } } while (0)

static napi_value fail(napi_env env, const char* what, const char* detail) {
    char msg[768];
    snprintf(msg, sizeof msg, "%s%s%s", what, detail && detail[0] ? ": " : "", detail ? detail : "");
    napi_throw_error(env, NULL, msg);
    return NULL;
}
static void finalize_handle(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    handle_t* h = (handle_t*)data;
    if (!h) return;
    if (h->ctx && h->ctx_destroy) h->ctx_destroy(h->ctx);
    if (h->lib) dlclose(h->lib);
    free(h);
}
static handle_t* get_handle(napi_env env, napi_value v) {
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) { napi_throw_type_error(env, NULL, "aero_worker: first argument must be a handle from open()"); return NULL; }
    return (handle_t*)p;
}
static int get_bytes(napi_env env, napi_value v, const uint8_t** data, size_t* len) {
    bool is_ta = false;
    if (napi_is_typedarray(env, v, &is_ta) == napi_ok && is_ta) {
        napi_typedarray_type type;
        void* p = NULL;
        size_t n = 0;
        if (napi_get_typedarray_info(env, v, &type, &n, &p, NULL, NULL) == napi_ok && type == napi_uint8_array) { *data = (const uint8_t*)p; *len = n; return 1; }
    }
    napi_throw_type_error(env, NULL, "aero_worker: expected a Uint8Array / Buffer");
    return 0;
}
static napi_value take_result(napi_env env, handle_t* h, uint8_t* buf, size_t n) {
    napi_value out;
    napi_status st = napi_create_buffer_copy(env, n, buf, NULL, &out);
    h->free_buf(buf);
    if (st != napi_ok) return fail(env, "aero_worker: could not allocate the result buffer", NULL);
    return out;
}

static napi_value Open(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 1) return fail(env, "aero_worker.open(libPath, device)", NULL);
    char path[1024];
    size_t plen = 0;
    CHECK(napi_get_value_string_utf8(env, argv[0], path, sizeof path, &plen));
    int32_t device = 0;
    if (argc > 1) CHECK(napi_get_value_int32(env, argv[1], &device));
    handle_t* h = (handle_t*)calloc(1, sizeof *h);
    if (!h) return fail(env, "aero_worker: out of memory", NULL);
    h->lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h->lib) { const char* e = dlerror(); free(h); return fail(env, "aero_worker: cannot load the library", e); }
#define BIND(field, name) do { *(void**)&h->field = dlsym(h->lib, name); if (!h->field) { dlclose(h->lib); free(h); return fail(env, "aero_worker: missing symbol", name); } } while (0)
    BIND(ctx_create, "aero_ctx_create"); BIND(ctx_destroy, "aero_ctx_destroy"); BIND(last_error, "aero_last_error"); BIND(free_buf, "aero_free");
    BIND(hash_rows, "aero_worker_hash_rows"); BIND(eval_constraints, "aero_worker_eval_constraints"); BIND(prover_output, "aero_prover_output");
#undef BIND
    if (device >= 0) {
        const int32_t rc = h->ctx_create(device, &h->ctx);
        if (rc != 0) {
            char why[512];
            snprintf(why, sizeof why, "status %d: %s", rc, h->last_error(NULL));
            dlclose(h->lib); free(h);
            return fail(env, "aero_worker: aero_ctx_create failed (no CPU fallback exists)", why);
        }
    }
    napi_value ext;
    if (napi_create_external(env, h, finalize_handle, NULL, &ext) != napi_ok) { finalize_handle(env, h, NULL); return fail(env, "aero_worker: could not create the handle", NULL); }
    return ext;
}
static napi_value Close(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t* h = argc ? get_handle(env, argv[0]) : NULL;
    if (!h) return NULL;
    if (h->ctx) { h->ctx_destroy(h->ctx); h->ctx = NULL; }      /* the library itself stays mapped until the handle is collected */
    return NULL;
}
static napi_value HashRows(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 2) return fail(env, "aero_worker.hashRows(handle, payload)", NULL);
    handle_t* h = get_handle(env, argv[0]);
    if (!h) return NULL;
    if (!h->ctx) return fail(env, "aero_worker.hashRows: the handle has no GPU context", NULL);
    const uint8_t* p; size_t n;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    uint8_t* out = NULL; size_t out_len = 0;
    const int32_t rc = h->hash_rows(h->ctx, p, n, &out, &out_len);
    if (rc != 0) return fail(env, "aero_worker_hash_rows", h->last_error(h->ctx));
    return take_result(env, h, out, out_len);
}
static napi_value EvalConstraints(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 2) return fail(env, "aero_worker.evalConstraints(handle, payload, air)", NULL);
    handle_t* h = get_handle(env, argv[0]);
    if (!h) return NULL;
    if (!h->ctx) return fail(env, "aero_worker.evalConstraints: the handle has no GPU context", NULL);
    const uint8_t* p; size_t n;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    aero_fib_air air = {0, 0, 0};
    const aero_fib_air* airp = NULL;
    if (argc > 2) {
        bool is_arr = false;
        CHECK(napi_is_array(env, argv[2], &is_arr));
        if (is_arr) {
            uint32_t* f[3] = {&air.aux_width, &air.aux_rands, &air.aux_degree};
            for (uint32_t i = 0; i < 3; i++) { napi_value e; CHECK(napi_get_element(env, argv[2], i, &e)); CHECK(napi_get_value_uint32(env, e, f[i])); }
            if (air.aux_width) airp = &air;
        }
    }
    uint8_t* out = NULL; size_t out_len = 0;
    const int32_t rc = h->eval_constraints(h->ctx, p, n, airp, &out, &out_len);
    if (rc != 0) return fail(env, "aero_worker_eval_constraints", h->last_error(h->ctx));
    return take_result(env, h, out, out_len);
}
static napi_value ProverOutput(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 3) return fail(env, "aero_worker.proverOutput(handle, proof, inputs)", NULL);
    handle_t* h = get_handle(env, argv[0]);
    if (!h) return NULL;
    const uint8_t *pp, *ip; size_t pn, in;
    if (!get_bytes(env, argv[1], &pp, &pn) || !get_bytes(env, argv[2], &ip, &in)) return NULL;
    uint8_t* out = NULL; size_t out_len = 0;
    char err[512] = {0};
    const int32_t rc = h->prover_output(pp, pn, ip, in, &out, &out_len, err, sizeof err);
    if (rc != 0) return fail(env, "aero_prover_output", err);
    return take_result(env, h, out, out_len);
}

NAPI_MODULE_INIT() {
    const struct { const char* name; napi_callback fn; } fns[] = {
        {"open", Open}, {"close", Close}, {"hashRows", HashRows}, {"evalConstraints", EvalConstraints}, {"proverOutput", ProverOutput}};
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
            napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) {
            napi_throw_error(env, NULL, "aero_worker: could not export its functions");
            return NULL;
        }
    }
    return exports;
}
