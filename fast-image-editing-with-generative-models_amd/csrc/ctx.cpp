// Context + error plumbing of the C ABI (include/fie.h).
#include <stdarg.h>
#include <string.h>
#include <iterator>
#include "fie_internal.h"

int fie_gemm_init(void);
int fie_gemm8_init(void);
int fie_gemm_w8_init(void);
int fie_attn_init(void);
int fie_gemm_x8_init(void);
int fie_conv_halo_init(void);

static thread_local char g_err[512] = "";

void fie_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void fie_oplog_append(fie_ctx* ctx, const void* fn, dim3 grid, dim3 block, unsigned lds) {
    const char* name = hipKernelNameRefByPtr(fn, ctx->stream);
    char line[512];
    snprintf(line, sizeof(line), "%s|%u|%u|%u|%s", name ? name : "?", grid.x * grid.y * grid.z, block.x * block.y * block.z, lds, ctx->op_desc);
    ctx->oplog->push_back(line);
    ctx->op_desc[0] = 0;
}

extern "C" {

int fie_version(void) { return 100; }

// Launch log for the per-shape profile (tools/shape_profile.py): while on, every launch of the library appends
// "kernel symbol|blocks|threads|dynamic LDS|description" (GEMM / conv / attention / norm ops describe their problem: shape, tile code,
// algorithmic FLOPs or bytes).  fie_debug_oplog_read copies the newline-joined log and returns its length (call with cap 0 to size).
int fie_debug_oplog(fie_ctx* ctx, int on) {
    FIE_REQUIRE(ctx != nullptr, "fie_debug_oplog: ctx is NULL");
    delete ctx->oplog;
    ctx->oplog = on ? new std::vector<std::string>() : nullptr;
    ctx->op_desc[0] = 0;
    return FIE_OK;
}

int fie_debug_oplog_mark(fie_ctx* ctx, const char* text) {      // a "#text" line between launches (stage boundaries)
    FIE_REQUIRE(ctx && text, "fie_debug_oplog_mark: NULL argument");
    if (ctx->oplog) ctx->oplog->push_back(std::string("#") + text);
    return FIE_OK;
}

int64_t fie_debug_oplog_read(fie_ctx* ctx, char* buf, int64_t cap) {
    if (!ctx || !ctx->oplog) return 0;
    int64_t n = 0;
    for (const std::string& l : *ctx->oplog) n += (int64_t)l.size() + 1;
    if (buf && cap > n) {
        char* q = buf;
        for (const std::string& l : *ctx->oplog) { memcpy(q, l.data(), l.size()); q += l.size(); *q++ = '\n'; }
        *q = 0;
    }
    return n;
}

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
    if (rc == FIE_OK) rc = fie_attn_init();
    if (rc == FIE_OK) rc = fie_gemm_x8_init();
    if (rc == FIE_OK) rc = fie_conv_halo_init();
    (void)hipSetDevice(cur);
    if (rc != FIE_OK) return rc;
    fie_ctx* c = new fie_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    c->num_cus = prop.multiProcessorCount;
    *out = c;
    return FIE_OK;
}

// ---- launch programs + graph-level entries (fie_unet_forward, fie_controlnet_forward, fie_vae_{encode,decode},
// fie_clip_text_forward): the host walks a graph ONCE with its static device buffers while a program records every launch;
// afterwards the named entry re-issues the whole graph from C++ (no Python, no shape logic, hipGraph-capturable like any launch).
int fie_program_begin(fie_ctx* ctx, fie_program** out) {
    FIE_REQUIRE(ctx && out, "fie_program_begin: NULL argument");
    FIE_REQUIRE(ctx->recording == nullptr, "fie_program_begin: a program is already being recorded on this ctx");
    *out = ctx->recording = new fie_program();
    return FIE_OK;
}

// `done` marks the end of the program's latest pass on `stream` (not under stream capture: a captured event cannot order eager work)
static void program_mark_done(fie_program* p, hipStream_t stream) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
    if (!p->done && hipEventCreateWithFlags(&p->done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); p->done = nullptr; return; }
    if (hipEventRecord(p->done, stream) != hipSuccess) { (void)hipGetLastError(); return; }
    p->has_done = true;
    p->last_stream = stream;
}

int fie_program_end(fie_ctx* ctx) {
    FIE_REQUIRE(ctx && ctx->recording, "fie_program_end: nothing is being recorded");
    program_mark_done(ctx->recording, ctx->stream);         // the recording pass executed on this stream
    ctx->recording = nullptr;
    return FIE_OK;
}

int fie_program_launches(const fie_program* p) { return p ? (int)p->recs.size() : -1; }

int fie_program_run(fie_ctx* ctx, fie_program* p) {
    FIE_REQUIRE(ctx && p, "fie_program_run: NULL argument");
    FIE_REQUIRE(ctx->recording != p, "fie_program_run: the program is still being recorded");
    // A program must not be in flight twice (frozen buffers, frozen split-K workspace): a run on ANOTHER stream than its previous pass
    // first waits for that pass.  Under stream capture nothing is ordered here: the captured graph inherits the rule (include/fie.h).
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool eager = hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
        if (eager && p->has_done && p->last_stream != ctx->stream && hipStreamWaitEvent(ctx->stream, p->done, 0) != hipSuccess) {
            fie_set_error("fie_program_run: hipStreamWaitEvent failed: %s", hipGetErrorString(hipGetLastError()));
            return FIE_EHIP;
        }
    }
    void* argv[64];
    for (const fie_launch_rec& r : p->recs) {
        FIE_REQUIRE(r.offs.size() <= 64, "fie_program_run: too many kernel arguments");
        for (size_t i = 0; i < r.offs.size(); ++i) argv[i] = const_cast<unsigned char*>(r.blob.data()) + r.offs[i];
        const hipError_t e = hipLaunchKernel(r.fn, r.grid, r.block, argv, r.lds, ctx->stream);
        if (e != hipSuccess) {
            fie_set_error("fie_program_run: launch failed: %s", hipGetErrorString(e));
            return FIE_EHIP;
        }
        if (ctx->recording) ctx->recording->recs.push_back(r);       // programs nest: running one while recording another copies it in
    }
    program_mark_done(p, ctx->stream);
    return FIE_OK;
}

int fie_program_destroy(fie_ctx* ctx, fie_program* p) {
    if (ctx)
        for (auto it = ctx->graphs.begin(); it != ctx->graphs.end();)
            it = it->second == p ? ctx->graphs.erase(it) : std::next(it);
    if (p && p->done) (void)hipEventDestroy(p->done);
    delete p;
    return FIE_OK;
}

int fie_graph_register(fie_ctx* ctx, const char* name, fie_program* p) {
    FIE_REQUIRE(ctx && name && p, "fie_graph_register: NULL argument");
    ctx->graphs[name] = p;
    return FIE_OK;
}

static int run_graph(fie_ctx* ctx, const char* name) {
    FIE_REQUIRE(ctx != nullptr, "%s: ctx is NULL", name);
    auto it = ctx->graphs.find(name);
    FIE_REQUIRE(it != ctx->graphs.end(), "%s: no program registered under this name (fie_graph_register)", name);
    return fie_program_run(ctx, it->second);
}

int fie_unet_forward(fie_ctx* ctx) { return run_graph(ctx, "unet_forward"); }
int fie_controlnet_forward(fie_ctx* ctx) { return run_graph(ctx, "controlnet_forward"); }
int fie_vae_encode(fie_ctx* ctx) { return run_graph(ctx, "vae_encode"); }
int fie_vae_decode(fie_ctx* ctx) { return run_graph(ctx, "vae_decode"); }
int fie_clip_text_forward(fie_ctx* ctx) { return run_graph(ctx, "clip_text_forward"); }

int fie_ctx_error_flag(fie_ctx* ctx, void* device_word) {
    FIE_REQUIRE(ctx != nullptr, "fie_ctx_error_flag: ctx is NULL");
    ctx->err_flag = static_cast<unsigned*>(device_word);
    return FIE_OK;
}

int fie_ctx_set_stream(fie_ctx* ctx, void* stream) {
    FIE_REQUIRE(ctx != nullptr, "fie_ctx_set_stream: ctx is NULL");
    ctx->stream = (hipStream_t)stream;
    return FIE_OK;
}

int fie_ctx_destroy(fie_ctx* ctx) {
    if (ctx && ctx->tune_buf) (void)hipFree(ctx->tune_buf);
    if (ctx && ctx->tune_flush) (void)hipFree(ctx->tune_flush);
    if (ctx) delete ctx->oplog;
    delete ctx;                       // registered programs are owned by the caller (fie_program_destroy)
    return FIE_OK;
}

}  // extern "C"
