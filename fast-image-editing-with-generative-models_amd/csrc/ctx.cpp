// Context + error plumbing of the C ABI (include/fie.h).
#include <stdarg.h>
#include <string.h>
#include "fie_internal.h"

int fie_gemm_init(void);
int fie_gemm8_init(void);
int fie_gemm_w8_init(void);

static thread_local char g_err[512] = "";

void fie_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int fie_version(void) { return 100; }

const char* fie_last_error(void) { return g_err; }

int fie_ctx_create(int device, void* stream, fie_ctx** out) {
    FIE_REQUIRE(out != nullptr, "fie_ctx_create: out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        fie_set_error("fie_ctx_create: device %d not available (%d visible)", device, n);
        return FIE_ENODEV;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        fie_set_error("fie_ctx_create: hipGetDeviceProperties failed");
        return FIE_EHIP;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fie_set_error("fie_ctx_create: device %d is %s; this library is built for gfx950 only", device,
                      prop.gcnArchName);
        return FIE_ENODEV;
    }
    // per-device kernel attributes (dynamic LDS sizes) are set here, once, not on the launch path
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (hipSetDevice(device) != hipSuccess) {
        fie_set_error("fie_ctx_create: hipSetDevice(%d) failed", device);
        return FIE_EHIP;
    }
    int rc = fie_gemm_init();
    if (rc == FIE_OK) rc = fie_gemm8_init();
    if (rc == FIE_OK) rc = fie_gemm_w8_init();
    (void)hipSetDevice(cur);
    if (rc != FIE_OK) return rc;
    fie_ctx* c = new fie_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    c->num_cus = prop.multiProcessorCount;
    *out = c;
    return FIE_OK;
}

int fie_ctx_set_stream(fie_ctx* ctx, void* stream) {
    FIE_REQUIRE(ctx != nullptr, "fie_ctx_set_stream: ctx is NULL");
    ctx->stream = (hipStream_t)stream;
    return FIE_OK;
}

int fie_ctx_destroy(fie_ctx* ctx) {
    delete ctx;
    return FIE_OK;
}

}  // extern "C"
