// Internal helpers shared by the HIP translation units of libfie_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <string>
#include <tuple>
#include <utility>
#include <vector>
#include "../../include/fie.h"

typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- launch programs (include/fie.h, fie_program_*): every kernel launch of the library goes through fie_launch(); while a
// program is being recorded the launch (kernel, grid, block, dynamic LDS, argument bytes) is appended to it, and
// fie_program_run() re-issues the list with hipLaunchKernel -- the graph-level entries (fie_unet_forward, ...) run such lists
// without any host-side shape logic.
struct fie_launch_rec {
    const void* fn;
    dim3 grid, block;
    unsigned lds;
    std::vector<unsigned char> blob;      // argument values, naturally aligned
    std::vector<unsigned> offs;           // offset of each argument in blob
};
struct fie_program {
    std::vector<fie_launch_rec> recs;
    // ordering of a program against ITSELF (round 4): its launches carry frozen pointers (static buffers, the split-K workspace that was
    // bound while it was recorded), so two copies in flight on different streams would race on them.  `done` is recorded behind the
    // recording pass and behind every run; a run on another stream than the previous one waits for it first (ctx.cpp: fie_program_run)
    hipEvent_t done = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_done = false;
};

struct fie_weight { const void* ptr; int64_t n, ld; };    // fie_weights_register: a packed device tensor under its diffusers parameter name
struct fie_step_cache { char* ptr; int64_t bytes; bool filled; };     // fie_step_cache_bind: what a model keeps over the denoising steps of one image

struct fie_tile_override { int mode, M, N, K, code; };   // tuning hook: per-shape GEMM / conv tile code (mode 0 GEMM, 1 conv)

struct fie_tune_key {      // one GEMM / conv problem as the autotuner sees it
    int mode, M, N, K, K1, geom, w8;
    bool operator<(const fie_tune_key& o) const {
        return std::tie(mode, M, N, K, K1, geom, w8) < std::tie(o.mode, o.M, o.N, o.K, o.K1, o.geom, o.w8);
    }
};

struct fie_ctx {
    int device;
    hipStream_t stream;
    int num_cus;
    // tuning / test hooks of the GEMM launch table (fie_debug_*): per ctx, never process-global
    int force_tile = 0;
    int gemm_probe = 0;
    int epi_prefetch = 1;                       // ring kernels load the epilogue's bias / residual operands before the K loop (A/B switch: fie_debug_epilogue_prefetch)
    unsigned* gemm_stamps = nullptr;
    float* gn_target = nullptr;              // fie_gn_stats_target: consumed by the next GEMM / conv launch
    int64_t gn_target_rows = 0;
    int gn_target_groups = 0;
    int attn_variant = 0;                    // fie_debug_attn_variant: 0 = v2 (LDS-DMA ring, deferred rescale), 1 = v1, 2 / 3 = v2 with 128 / 64 queries per block forced
    int gn_onepass = 1;                      // fie_debug_gn_onepass: 0 = always the three-kernel GroupNorm
    unsigned* err_flag = nullptr;            // fie_ctx_error_flag: device word kernels set to a FIE_DEVERR_* code instead of failing silently
    void* sk_ws = nullptr;                   // fie_splitk_workspace: [4096 arrival counters (zero between launches)][fp32 partial-tile slabs]
    int64_t sk_bytes = 0;
    int splitk_mode = 1;                     // fie_debug_splitk: 0 = never split K, 1 = where the tuner / an override says so
    int autotune = 0;                        // fie_gemm_autotune: time the eligible tiles at a shape's first eager launch
    std::map<fie_tune_key, int> tuned;
    int tune_exclude[16] = {0};              // fie_debug_tune_exclude: tile codes the tuner must not offer (10000 = every split-K variant); 0-terminated
    void* tune_buf = nullptr;                // scratch output of the timing launches
    void* tune_flush = nullptr;              // 384 MB written between timed launches: cold weights, as inside the network
    size_t tune_bytes = 0;
    int n_overrides = 0;
    fie_tile_override overrides[32];
    char last_kernel[128] = "";
    fie_program* recording = nullptr;                 // launches are appended here while set (fie_program_begin / _end)
    std::vector<std::string>* oplog = nullptr;        // fie_debug_oplog: one line per launch (kernel symbol, grid, block, LDS, the op's own description)
    char op_desc[192] = "";                           // FIE_DESC: description of the op whose next launch is logged (consumed by that launch)
    std::map<std::string, fie_program*> graphs;       // fie_graph_register: "unet_forward", "vae_decode", ...
    std::map<std::string, fie_step_cache> step_caches;   // by model prefix (graphs.cpp)
    std::map<std::string, fie_weight> weights;        // fie_weights_register: what the C++ graph walks (graphs.cpp: fie_vae_decode_f16) look up
};

template <typename T>
inline void fie_pack_arg(fie_launch_rec& r, const T& v) {
    size_t off = (r.blob.size() + alignof(T) - 1) / alignof(T) * alignof(T);
    r.blob.resize(off + sizeof(T));
    memcpy(r.blob.data() + off, &v, sizeof(T));
    r.offs.push_back((unsigned)off);
}

// fie_debug_oplog (tools/shape_profile.py pairs the lines with a rocprofv3 kernel trace): ops describe themselves only while a log is open
void fie_oplog_append(fie_ctx* ctx, const void* fn, dim3 grid, dim3 block, unsigned lds);
#define FIE_DESC(ctx, ...)                                                              \
    do {                                                                                \
        if ((ctx)->oplog) snprintf((ctx)->op_desc, sizeof((ctx)->op_desc), __VA_ARGS__); \
    } while (0)

// The one launch point of the library: asynchronous on the ctx stream; recorded when a program is open.
template <typename... KArgs, typename... Args>
inline void fie_launch(fie_ctx* ctx, void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned lds, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count");
    std::tuple<KArgs...> vals(static_cast<KArgs>(args)...);
    std::apply([&](const KArgs&... a) { hipLaunchKernelGGL(kernel, grid, block, lds, ctx->stream, a...); }, vals);
    if (ctx->oplog) fie_oplog_append(ctx, reinterpret_cast<const void*>(kernel), grid, block, lds);
    if (ctx->recording) {
        fie_launch_rec r;
        r.fn = reinterpret_cast<const void*>(kernel);
        r.grid = grid; r.block = block; r.lds = lds;
        std::apply([&](const KArgs&... a) { (fie_pack_arg(r, a), ...); }, vals);
        ctx->recording->recs.push_back(std::move(r));
    }
}

void fie_set_error(const char* fmt, ...);

#define FIE_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) {                                 \
            fie_set_error(__VA_ARGS__);                \
            return FIE_EINVAL;                         \
        }                                              \
    } while (0)

#define FIE_LAUNCH_CHECK()                                                         \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            fie_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,           \
                          hipGetErrorString(e__));                                 \
            return FIE_EHIP;                                                       \
        }                                                                          \
    } while (0)

static inline int64_t fie_roundup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// value -> the e4m3 range, saturating: +-448 for finite inputs; a NaN stays a NaN and +-Inf becomes one (the conversion writes the e4m3 NaN code and
// the consuming MFMA spreads it), so a numerical blow-up upstream stays as visible in the fp8 configuration as the fp16 path would leave it
__device__ __forceinline__ float fie_sat448(float x) {
    const float c = fminf(fmaxf(x, -448.f), 448.f);
    return fabsf(x) <= 3.402823466e38f ? c : __builtin_nanf("");
}
__device__ __forceinline__ float fie_silu(float x) { return x / (1.0f + __expf(-x)); }
// exact-erf GELU.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, three orders below fp16 resolution): one rcp, one
// exp and five FMAs instead of libm's branchy erff -- the GEGLU epilogue evaluates it for every FF1 output element.
__device__ __forceinline__ float fie_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float y = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    return copysignf(1.0f - y * __expf(-ax * ax), x);
}
__device__ __forceinline__ float fie_gelu(float x) { return 0.5f * x * (1.0f + fie_erf(x * 0.70710678118654752f)); }
__device__ __forceinline__ float fie_qgelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }

__device__ __forceinline__ float fie_gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// 8 consecutive elements <-> float[8] for either storage type (fp16: one 16-byte access, fp32: two)
__device__ __forceinline__ void fie_load8(const half_t* p, float (&x)[8]) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)v[j];
}
__device__ __forceinline__ void fie_load8(const float* p, float (&x)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}
__device__ __forceinline__ void fie_store8(half_t* p, const float (&x)[8]) {
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (half_t)x[j];
    *reinterpret_cast<f16x8*>(p) = v;
}
__device__ __forceinline__ void fie_store8(float* p, const float (&x)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(x[4], x[5], x[6], x[7]);
}
