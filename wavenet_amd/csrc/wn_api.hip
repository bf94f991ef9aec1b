// C ABI of libwavenet_hip.so (include/wavenet_hip.h): context, filter state, kernel dispatch,
// batch reduction, on-device update, RCCL exchange.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "wavenet_hip.h"
#include "wn_costfn.h"
#include "wn_internal.h"
#include "wn_kernels.h"

// kernel tables, one translation unit per compiled tap count (wn_fused_np.hip)
extern "C" const void* wn_kernel_np2(int kind, int sc);
extern "C" const void* wn_kernel_np4(int kind, int sc);
extern "C" const void* wn_kernel_np8(int kind, int sc);
extern "C" const void* wn_kernel_np16(int kind, int sc);
extern "C" const void* wn_kernel_np32(int kind, int sc);
extern "C" const void* wn_kernel_wave64(int np);
extern "C" const void* wn_kernel_wave64_transform(int kind, int np);
extern "C" const void* wn_kernel_wave64_sparse();
extern "C" const void* wn_kernel_wave1d(int np);
extern "C" const void* wn_kernel_wave1d_line(int np, int M);
extern "C" const void* wn_kernel_sparse(int sc);

namespace {

enum { KIND_TRAIN = 0, KIND_FORWARD = 1, KIND_INVERSE = 2, KIND_BACKPROP = 3 };

struct SizeClass { int T, Q, NS; };
const SizeClass kSizeClasses[5] = {{64, 1, 128}, {64, 3, 512}, {64, 4, 1024}, {256, 3, 2048}, {512, 3, 4096}};
const int kLogCap = 16384;

}  // namespace

// Partial sums of wn_train_batch_static: a fixed array in the code object instead of memory from hipMalloc,
// for callers that cannot allocate.  Every GPU the library runs on holds its own copy at its own address;
// scratch_ptr[device] is that address, looked up once per device in wn_create.  In host code the array's
// name is only a handle for that look-up, never a device address.
// Every context of a GPU shares that GPU's array.  scratch_last[device] is the context whose call used it
// last; a call from another context first makes its stream wait for that context's static_done event, so
// that calls issued one after the other never overlap on the array whatever streams they run on.
// scratch_mutex guards the look-up, scratch_last and the wait/launch/record sequence against host threads.
constexpr int kStaticSlots = WN_STATIC_SLOTS;
static __device__ __align__(256) double scratch[(size_t)kStaticSlots * wn::PSTRIDE];
constexpr int kMaxDevices = 64;
static std::mutex scratch_mutex;
static double* scratch_ptr[kMaxDevices];
static wn_context* scratch_last[kMaxDevices];

namespace {

using wn::fail;

int ilog2_exact(int v) {  // -1 unless v is a power of two >= 1
    if (v < 1 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) l++;
    return l;
}

int size_class(long long n) {
    if (n <= 128) return 0;
    if (n <= 512) return 1;
    if (n <= 1024) return 2;
    if (n <= 2048) return 3;
    if (n <= 4096) return 4;
    return -1;
}

int compiled_taps(int N) {
    for (int np : {2, 4, 8, 16, 32}) {
        if (N <= np) return np;
    }
    return -1;
}

const void* fused_kernel(int NP, int kind, int sc) {
    switch (NP) {
        case 2: return wn_kernel_np2(kind, sc);
        case 4: return wn_kernel_np4(kind, sc);
        case 8: return wn_kernel_np8(kind, sc);
        case 16: return wn_kernel_np16(kind, sc);
        case 32: return wn_kernel_np32(kind, sc);
    }
    return nullptr;
}

// Checks (B, R, C) and the filter; fills lgR, lgC, sc.
int check_shape(wn_context* ctx, int B, int R, int C, bool need_filter, int* lgR, int* lgC, int* sc) {
    if (!ctx) return WN_ERR_ARG;
    if (B < 0) return fail(ctx, WN_ERR_ARG, "negative batch size");
    *lgR = ilog2_exact(R);
    *lgC = ilog2_exact(C);
    if (*lgR < 0 || *lgC < 0)
        return fail(ctx, WN_ERR_BAD_SHAPE, "shape {" + std::to_string(R) + ", " + std::to_string(C) + "} is not radix 2");
    if (need_filter && !ctx->has_filter) return fail(ctx, WN_ERR_STATE, "no filter set");
    *sc = size_class((long long)R * C);  // -1: beyond the LDS-resident kernels, streamed path
    if (*lgR + *lgC > 30) return fail(ctx, WN_ERR_UNSUPPORTED, "more than 2^30 samples per example");
    return WN_OK;
}

int launch(wn_context* ctx, const void* func, int T, size_t shmem, int B, wn::FusedArgs& A, int* grid_out, int grid_cap = 1 << 30) {
    if (!func) return fail(ctx, WN_ERR_UNSUPPORTED, "no kernel compiled for this configuration");
    auto it = ctx->lds_configured.find(func);
    if (it == ctx->lds_configured.end() || it->second < shmem) {
        WN_HIP(ctx, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        ctx->lds_configured[func] = shmem;
    }
    auto key = std::make_pair(func, shmem);
    auto oc = ctx->occupancy.find(key);
    int perCU;
    if (oc == ctx->occupancy.end()) {
        int nb = 0;
        WN_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, func, T, shmem));
        perCU = nb < 1 ? 1 : nb;
        ctx->occupancy[key] = perCU;
    } else {
        perCU = oc->second;
    }
    long long grid = (long long)perCU * ctx->numCU;
    if (grid > B) grid = B;
    if (grid > grid_cap) grid = grid_cap;   // the kernels stride over the batch: fewer workgroups only take more turns
    if (grid < 1) grid = 1;
    if (grid_out) *grid_out = (int)grid;
    void* args[] = {&A};
    WN_HIP(ctx, hipLaunchKernel(func, dim3((unsigned)grid), dim3((unsigned)T), args, shmem, ctx->stream));
    return WN_OK;
}

// ---- small kernels ----------------------------------------------------------------------------

// Sums the partial-sum slots: sums[0..N) gradient, [N] sparse term, [N+1] non-finite count.  One
// wave per quantity; lane l adds slots l, l + 64, ... in order, then the 64 lane sums are combined by
// a butterfly -- a fixed order, so the result does not depend on timing.
__global__ __launch_bounds__(64) void k_reduce_partials(const double* partial, int nslots, int N, double* sums) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t > N + 1) return;
    const double s = wn::reduce_slots(partial, nslots, wn::PSTRIDE, t, lane);
    if (lane == 0) sums[t] = s;
}

// Wavenet::flushBatchQueue_ / update_ / setFilter (src/Wavenet.cxx:717-765, 8-36) as a launch of its own:
// after an all-reduce over ranks, and for examples that take the streamed path (wn_step.h has the body).
__global__ __launch_bounds__(64) void k_update(wn::FilterDev* f, const double* sums, wn::UpdateParams P) {
    __shared__ wn::UpdateShared sh;
    const wn::UpdateInputs state(f, (int)threadIdx.x);
    wn::update_body(f, state, sums, P, sh, (int)threadIdx.x);
}

// Batch reduction and update in one launch, for the training kernels whose many small workgroups leave too
// many partial-sum slots for one workgroup's tail: wave w adds up quantities w, w + 16, ... (the order of
// k_reduce_partials), then the first wave updates the filter (mode 2) -- or the sums are left for the
// caller's all-reduce (mode 1).
__global__ __launch_bounds__(1024) void k_finish(const double* partial, int nslots, wn::StepTail S) {
    __shared__ double ssum[wn::PSTRIDE];
    __shared__ double lsum[wn::PSTRIDE][64];
    __shared__ wn::UpdateShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const wn::UpdateInputs state(S.filt, tid);     // on its way while the slots are added up
    const int N = state.N, nq = N + 2;
    // The sums in the order of reduce_slots (slots l, l + 64, ... added up in order, then the 64 partial results by
    // a shuffle tree), computed from whole rows: a thread owns (l, quantity) pairs with consecutive threads on
    // consecutive quantities of a slot, so the slots are read as rows, not as one cache line per lane and quantity.
    for (int pair = tid; pair < 64 * nq; pair += 1024) {
        const int l = pair / nq, q = pair - l * nq;
        double s = 0;
        for (int base = l; base < nslots; base += 64 * 16) {
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int slot = base + 64 * i;
                v[i] = (slot < nslots) ? partial[(size_t)slot * wn::PSTRIDE + q] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (base + 64 * i < nslots) s += v[i];
            }
        }
        lsum[q][l] = s;
    }
    __syncthreads();
    for (int q = wave; q < nq; q += 16) {
        const double s = wn::wave_sum(lsum[q][lane]);
        if (lane == 0) { ssum[q] = s; S.sums[q] = s; }
    }
    __syncthreads();
    if (S.mode == 2) wn::update_body(S.filt, state, ssum, S.P, sh, tid);
}

// The update state in one piece for a single copy to the host: out[0] = number of updates, out[1] = non-finite flag,
// out[2 .. 2 + MAXTAPS) = momentum, then for the updates first .. min(first + max_count, updates) - 1 rows of
// MAXTAPS + 1 doubles: the filter after the update and the batch's mean cost.
__global__ __launch_bounds__(256) void k_pack_updates(const wn::FilterDev* f, const double* logF, const double* logC, int first, int max_count,
                                                      int cap, double* out) {
    const int steps = f->steps;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) { out[0] = (double)steps; out[1] = (double)f->nonfinite; }
        if (threadIdx.x < wn::MAXTAPS) out[2 + threadIdx.x] = f->momentum[threadIdx.x];
    }
    const int count = min(max_count, steps - first);
    constexpr int ROW = wn::MAXTAPS + 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count * ROW; i += gridDim.x * blockDim.x) {
        const int r = i / ROW, c = i % ROW, idx = (first + r) % cap;
        out[2 + wn::MAXTAPS + i] = (c < wn::MAXTAPS) ? logF[(size_t)idx * wn::MAXTAPS + c] : logC[idx];
    }
}

__global__ void k_reset_steps(wn::FilterDev* f) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { f->steps = 0; f->nonfinite = 0; }
}

// ---- RCCL through dlopen: the library is only needed by multi-GPU callers ----------------------
struct NcclId { char internal[128]; };
typedef int (*fn_ncclGetUniqueId)(NcclId*);
typedef int (*fn_ncclCommInitRank)(void**, int, NcclId, int);
typedef int (*fn_ncclAllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_ncclCommDestroy)(void*);
typedef const char* (*fn_ncclGetErrorString)(int);
const int kNcclDouble = 8, kNcclSum = 0;

void* open_rccl() {
    static void* lib = nullptr;
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    return lib;
}

}  // namespace

extern "C" {

const char* wn_status_string(int status) {
    switch (status) {
        case WN_OK: return "ok";
        case WN_ERR_ARG: return "bad argument";
        case WN_ERR_BAD_SHAPE: return "shape is not radix 2";
        case WN_ERR_BAD_FILTER: return "filter is empty or of odd length";
        case WN_ERR_UNSUPPORTED: return "unsupported size";
        case WN_ERR_HIP: return "HIP runtime error";
        case WN_ERR_COMM: return "RCCL error";
        case WN_ERR_NONFINITE: return "non-finite cost or gradient";
        case WN_ERR_STATE: return "call order";
    }
    return "unknown status";
}

const char* wn_last_error(const wn_context* ctx) { return ctx ? ctx->err.c_str() : wn::global_error().c_str(); }

int wn_create(int device, wn_context** out) {
    if (!out) return WN_ERR_ARG;
    *out = nullptr;
    wn_context* ctx = new wn_context();
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
        wn::global_error() = std::string("hipSetDevice: ") + hipGetErrorString(e);
        delete ctx;
        return WN_ERR_HIP;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        wn::global_error() = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        delete ctx;
        return WN_ERR_HIP;
    }
    ctx->numCU = prop.multiProcessorCount;
    if (device < 0 || device >= kMaxDevices) {
        wn::global_error() = "device number beyond " + std::to_string(kMaxDevices - 1);
        delete ctx;
        return WN_ERR_ARG;
    }
    {
        std::lock_guard<std::mutex> lock(scratch_mutex);
        if (!scratch_ptr[device]) {   // this device is current: the address is the one of its copy
            double* p = nullptr;
            e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !p) {
                wn::global_error() = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
                delete ctx;
                return WN_ERR_HIP;
            }
            scratch_ptr[device] = p;
        }
    }
    e = hipEventCreateWithFlags(&ctx->static_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        wn::global_error() = std::string("hipEventCreateWithFlags: ") + hipGetErrorString(e);
        delete ctx;
        return WN_ERR_HIP;
    }
    int rc = WN_OK;
    auto alloc = [&](void** p, size_t bytes) {
        hipError_t ee = hipMalloc(p, bytes);
        if (ee == hipSuccess) ee = hipMemset(*p, 0, bytes);
        if (ee != hipSuccess) { wn::global_error() = std::string("hipMalloc: ") + hipGetErrorString(ee); rc = WN_ERR_HIP; }
    };
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) rc = WN_ERR_HIP;
    ctx->own_stream = true;
    ctx->partial_blocks = ctx->numCU * 32;
    if (const char* w = getenv("WAVENET_HIP_WAVE64")) ctx->wave64 = (w[0] != '0');
    if (const char* w = getenv("WAVENET_HIP_EXACT_SORT")) ctx->exact_sort = (w[0] == '1');
    if (const char* w = getenv("WAVENET_HIP_RANK")) ctx->bucket_rank = (strcmp(w, "radix") != 0);
    if (const char* w = getenv("WAVENET_HIP_ROW_LINE_TAPS")) ctx->row_line_taps = std::min(16, std::max(0, atoi(w)));
    if (const char* w = getenv("WAVENET_HIP_FUSED_HIST")) ctx->fused_hist = (strcmp(w, "0") != 0);
    if (const char* w = getenv("WAVENET_HIP_RANK_PAIRS")) ctx->rank_pairs_cols = (strcmp(w, "0") != 0);
    if (const char* w = getenv("WAVENET_HIP_RANK_PAIRS_TAPS")) ctx->rank_pairs_taps = atoi(w);
    if (const char* w = getenv("WAVENET_HIP_RANK_REGIONS_FROM")) ctx->rank_regions_from = std::min(64, std::max(0, atoi(w)));
    if (const char* w = getenv("WAVENET_HIP_RANK_REGIONS_IB")) ctx->rank_regions_ib = std::min(15, std::max(13, atoi(w)));
    if (const char* w = getenv("WAVENET_HIP_RANK_KEYS")) ctx->rank_packed = (strcmp(w, "12") != 0);
    if (const char* w = getenv("WAVENET_HIP_RANK_SUB")) ctx->rank_sub = std::max(0, atoi(w));
    if (const char* w = getenv("WAVENET_HIP_CHUNK")) ctx->chunk_cap = std::max(0, atoi(w));
    if (rc == WN_OK) alloc((void**)&ctx->dfilt, sizeof(wn::FilterDev));
    if (rc == WN_OK) alloc((void**)&ctx->partial, (size_t)ctx->partial_blocks * wn::PSTRIDE * sizeof(double));
    if (rc == WN_OK) alloc((void**)&ctx->own_sums, wn::PSTRIDE * sizeof(double));
    if (rc == WN_OK) alloc((void**)&ctx->ticket, 256);
    ctx->sums = ctx->own_sums;
    if (rc == WN_OK) alloc((void**)&ctx->log_filters, (size_t)kLogCap * wn::MAXTAPS * sizeof(double));
    if (rc == WN_OK) alloc((void**)&ctx->log_costs, (size_t)kLogCap * sizeof(double));
    if (rc != WN_OK) {
        wn_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return WN_OK;
}

int wn_destroy(wn_context* ctx) {
    wn::make_current(ctx);
    if (!ctx) return WN_OK;
    (void)hipSetDevice(ctx->device);
    wn_comm_destroy(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->device >= 0 && ctx->device < kMaxDevices) {
        std::lock_guard<std::mutex> lock(scratch_mutex);
        if (scratch_last[ctx->device] == ctx) scratch_last[ctx->device] = nullptr;  // its calls have finished
    }
    if (ctx->static_done) (void)hipEventDestroy(ctx->static_done);
    (void)hipFree(ctx->dfilt);
    (void)hipFree(ctx->partial);
    (void)hipFree(ctx->own_sums);
    (void)hipFree(ctx->ticket);
    if (ctx->rank_event) (void)hipEventDestroy(ctx->rank_event);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    (void)hipFree(ctx->workspace);
    if (ctx->gen_tables) (void)hipFree(ctx->gen_tables);
    free(ctx->gen_tables_host);
    if (ctx->grid_filters) (void)hipFree(ctx->grid_filters);
    if (ctx->grid_scratch) (void)hipFree(ctx->grid_scratch);
    for (hipEvent_t ev : ctx->prof_begin) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->prof_end) (void)hipEventDestroy(ev);
    (void)hipFree(ctx->log_filters);
    (void)hipFree(ctx->log_costs);
    if (ctx->pack_dev) (void)hipFree(ctx->pack_dev);
    if (ctx->pack_host) (void)hipHostFree(ctx->pack_host);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return WN_OK;
}

int wn_set_stream(wn_context* ctx, void* hip_stream) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    WN_HIP(ctx, hipSetDevice(ctx->device));
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->own_stream = false;
    ctx->stream = (hipStream_t)hip_stream;  // NULL is HIP's default stream
    return WN_OK;
}

int wn_synchronize(wn_context* ctx) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WN_OK;
}

int wn_get_device(wn_context* ctx, int* device) {
    if (!ctx || !device) return WN_ERR_ARG;
    *device = ctx->device;
    return WN_OK;
}

int wn_malloc(wn_context* ctx, void** dev_ptr, size_t bytes) {
    wn::make_current(ctx);
    if (!ctx || !dev_ptr) return WN_ERR_ARG;
    WN_HIP(ctx, hipSetDevice(ctx->device));
    WN_HIP(ctx, hipMalloc(dev_ptr, bytes ? bytes : 8));
    return WN_OK;
}

int wn_free(wn_context* ctx, void* dev_ptr) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WN_HIP(ctx, hipFree(dev_ptr));
    return WN_OK;
}

int wn_memcpy_h2d(wn_context* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    wn::make_current(ctx);
    if (!ctx || (bytes && (!dst_dev || !src_host))) return WN_ERR_ARG;
    WN_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WN_OK;
}

int wn_memcpy_d2h(wn_context* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    wn::make_current(ctx);
    if (!ctx || (bytes && (!dst_host || !src_dev))) return WN_ERR_ARG;
    WN_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WN_OK;
}

// ---- filter ---------------------------------------------------------------------------------------

int wn_set_filter(wn_context* ctx, const double* a, int N) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    if (!a || N <= 0) return fail(ctx, WN_ERR_BAD_FILTER, "Input filter is empty.");
    if (N % 2) return fail(ctx, WN_ERR_BAD_FILTER, "Input filter length is not a multiple of 2.");
    const int NP = compiled_taps(N);
    if (NP < 0)
        return fail(ctx, WN_ERR_UNSUPPORTED, "filter length " + std::to_string(N) + " exceeds " + std::to_string(WN_MAX_TAPS));
    wn::FilterDev h;
    memset(&h, 0, sizeof(h));
    const int d = (NP - N) / 2;
    for (int j = 0; j < N; j++) {
        h.a_true[j] = a[j];
        h.a[j + d] = a[j];
        h.b[j + d] = wn::parity_sign(j) * a[N - 1 - j];
        h.vL[j + d] = 1.0;
        h.sH[j + d] = wn::parity_sign(N - 1 - j);
    }
    h.N = N;
    h.NP = NP;
    h.d = d;
    // taps and lengths; momentum and the log counter stay on the device
    WN_HIP(ctx, hipMemcpyAsync(ctx->dfilt, &h, offsetof(wn::FilterDev, momentum), hipMemcpyHostToDevice, ctx->stream));
    WN_HIP(ctx, hipMemcpyAsync(&ctx->dfilt->N, &h.N, 3 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->has_filter || ctx->N != N) {
        WN_HIP(ctx, hipMemsetAsync(ctx->dfilt->momentum, 0, sizeof(h.momentum), ctx->stream));  // :31-33
    }
    WN_HIP(ctx, hipMemsetAsync(&ctx->dfilt->nonfinite, 0, sizeof(int), ctx->stream));
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));  // h is a stack object
    ctx->has_filter = true;
    ctx->N = N;
    ctx->NP = NP;
    return WN_OK;
}

int wn_get_filter(wn_context* ctx, double* a, int* N) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    if (!ctx->has_filter) return fail(ctx, WN_ERR_STATE, "no filter set");
    if (N) *N = ctx->N;
    if (a) return wn_memcpy_d2h(ctx, a, ctx->dfilt->a_true, (size_t)ctx->N * sizeof(double));
    return WN_OK;
}

int wn_set_momentum(wn_context* ctx, const double* p, int N) {
    wn::make_current(ctx);
    if (!ctx || !p) return WN_ERR_ARG;
    if (!ctx->has_filter || N != ctx->N) return fail(ctx, WN_ERR_STATE, "momentum length differs from the filter's");
    return wn_memcpy_h2d(ctx, ctx->dfilt->momentum, p, (size_t)N * sizeof(double));
}

int wn_get_momentum(wn_context* ctx, double* p, int* N) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    if (!ctx->has_filter) return fail(ctx, WN_ERR_STATE, "no filter set");
    if (N) *N = ctx->N;
    if (p) return wn_memcpy_d2h(ctx, p, ctx->dfilt->momentum, (size_t)ctx->N * sizeof(double));
    return WN_OK;
}

static int dense_operator(const double* f, int N, unsigned size, double* M) {
    if (!f || !M || N <= 0) return WN_ERR_ARG;
    if (N % 2) return WN_ERR_BAD_FILTER;
    if (size > 13) return WN_ERR_UNSUPPORTED;  // 2^13 x 2^14 doubles = 1 GiB
    const size_t nRows = (size_t)1 << size, nCols = (size_t)2 << size;
    memset(M, 0, nRows * nCols * sizeof(double));
    for (size_t r = 0; r < nRows; r++) {
        for (int j = 0; j < N; j++) {
            const size_t c = (size_t)((long long)(N / 2) + 2 * (long long)r - j) & (nCols - 1);
            M[r + c * nRows] += f[j];
        }
    }
    return WN_OK;
}

int wn_lowpass_operator(const double* a, int N, unsigned size, double* M) { return dense_operator(a, N, size, M); }

int wn_highpass_operator(const double* a, int N, unsigned size, double* M) {
    if (!a || N <= 0 || N > 4096) return WN_ERR_ARG;
    if (N % 2) return WN_ERR_BAD_FILTER;
    double b[4096];
    for (int i = 0; i < N; i++) b[i] = wn::parity_sign(i) * a[N - 1 - i];
    return dense_operator(b, N, size, M);
}

// ---- transforms -----------------------------------------------------------------------------------

int wn_forward(wn_context* ctx, const double* X, int B, int R, int C, double* Y, double* Prow, double* Y1, double* Pcol) {
    wn::make_current(ctx);
    int lgR, lgC, sc;
    int rc = check_shape(ctx, B, R, C, true, &lgR, &lgC, &sc);
    if (rc) return rc;
    if (B == 0) return WN_OK;
    if (!X) return fail(ctx, WN_ERR_ARG, "null input");
    if (sc < 0) return wn::large_forward(ctx, X, B, lgR, lgC, Y, Prow, Y1, Pcol);
    const SizeClass& S = kSizeClasses[sc];
    wn::FusedArgs A;
    memset(&A, 0, sizeof(A));
    A.B = B; A.lgR = lgR; A.lgC = lgC; A.filt = ctx->dfilt;
    A.X = X; A.Y = Y; A.Prow = Prow; A.Y1 = Y1; A.Pcol = Pcol;
    if (const void* w64 = (lgR == 6 && lgC == 6 && ctx->wave64) ? wn_kernel_wave64_transform(KIND_FORWARD, ctx->NP) : nullptr)
        return launch(ctx, w64, 256, (size_t)4 * wn::W64_TILE * sizeof(double), (B + 3) / 4, A, nullptr);
    const size_t shmem = (size_t)4 * R * C * sizeof(double);
    return launch(ctx, fused_kernel(ctx->NP, KIND_FORWARD, sc), S.T, shmem, B, A, nullptr);
}

int wn_inverse(wn_context* ctx, const double* Y, int B, int R, int C, double* X) {
    wn::make_current(ctx);
    int lgR, lgC, sc;
    int rc = check_shape(ctx, B, R, C, true, &lgR, &lgC, &sc);
    if (rc) return rc;
    if (B == 0) return WN_OK;
    if (!Y || !X) return fail(ctx, WN_ERR_ARG, "null pointer");
    if (sc < 0) return wn::large_inverse(ctx, Y, B, lgR, lgC, X);
    const SizeClass& S = kSizeClasses[sc];
    wn::FusedArgs A;
    memset(&A, 0, sizeof(A));
    A.B = B; A.lgR = lgR; A.lgC = lgC; A.filt = ctx->dfilt;
    A.Yin = Y; A.Xout = X;
    if (const void* w64 = (lgR == 6 && lgC == 6 && ctx->wave64) ? wn_kernel_wave64_transform(KIND_INVERSE, ctx->NP) : nullptr)
        return launch(ctx, w64, 256, (size_t)4 * wn::W64_TILE * sizeof(double), (B + 3) / 4, A, nullptr);
    const size_t shmem = (size_t)R * C * sizeof(double) + 16;
    return launch(ctx, fused_kernel(ctx->NP, KIND_INVERSE, sc), S.T, shmem, B, A, nullptr);
}

int wn_backprop(wn_context* ctx, const double* Delta, const double* X, const double* Prow, const double* Y1,
                const double* Pcol, int B, int R, int C, double* grad, double* din) {
    wn::make_current(ctx);
    int lgR, lgC, sc;
    int rc = check_shape(ctx, B, R, C, true, &lgR, &lgC, &sc);
    if (rc) return rc;
    if (B == 0) return WN_OK;
    if (!Delta || !X || !Prow || !Y1 || !Pcol || !grad) return fail(ctx, WN_ERR_ARG, "null pointer");
    if (sc < 0) return wn::large_backprop(ctx, Delta, X, Prow, Y1, Pcol, B, lgR, lgC, grad, din);
    const SizeClass& S = kSizeClasses[sc];
    wn::FusedArgs A;
    memset(&A, 0, sizeof(A));
    A.B = B; A.lgR = lgR; A.lgC = lgC; A.filt = ctx->dfilt;
    A.Din = Delta; A.X = X; A.ProwIn = Prow; A.Y1In = Y1; A.PcolIn = Pcol; A.grad = grad; A.Delta = din;
    if (const void* w64 = (lgR == 6 && lgC == 6 && ctx->wave64) ? wn_kernel_wave64_transform(KIND_BACKPROP, ctx->NP) : nullptr)
        return launch(ctx, w64, 256, (size_t)4 * wn::W64_TILE * sizeof(double), (B + 3) / 4, A, nullptr);
    const size_t shmem = ((size_t)4 * R * C + (size_t)(S.T / 64) * ctx->NP + 2) * sizeof(double);
    return launch(ctx, fused_kernel(ctx->NP, KIND_BACKPROP, sc), S.T, shmem, B, A, nullptr);
}

// ---- cost functions -------------------------------------------------------------------------------

int wn_sparse_term_deriv(wn_context* ctx, const double* Y, int B, int n, double* Delta, double* Sout) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    if (B < 0 || n < 1) return fail(ctx, WN_ERR_ARG, "bad counts");
    // any n is accepted by the reference; the sort kernels take powers of two
    int lgn = ilog2_exact(n);
    if (lgn < 0) return fail(ctx, WN_ERR_BAD_SHAPE, "number of coefficients is not radix 2");
    const int sc = size_class(n);
    if (B == 0) return WN_OK;
    if (!Y) return fail(ctx, WN_ERR_ARG, "null input");
    if (sc < 0) return wn::large_sparse(ctx, Y, B, n, Delta, Sout);
    const SizeClass& S = kSizeClasses[sc];
    wn::FusedArgs A;
    memset(&A, 0, sizeof(A));
    A.B = B; A.lgR = lgn; A.lgC = 0;
    A.Yin = Y; A.Delta = Delta; A.S = Sout;
    if (n == 4096 && ctx->wave64)
        return launch(ctx, wn_kernel_wave64_sparse(), 256, (size_t)4 * wn::W64_TILE * sizeof(double), (B + 3) / 4, A, nullptr);
    const size_t shmem = ((size_t)S.NS + S.T / 64 + 2) * sizeof(double) + 16;
    return launch(ctx, wn_kernel_sparse(sc), S.T, shmem, B, A, nullptr);
}

int wn_sparse_term(wn_context* ctx, const double* Y, int B, int n, double* Sout) {
    wn::make_current(ctx);
    if (ctx && !Sout) return fail(ctx, WN_ERR_ARG, "null output");
    return wn_sparse_term_deriv(ctx, Y, B, n, nullptr, Sout);
}

int wn_reg_term(const double* a, int N, int do_wavelet, double* R_out) {
    if (!a || !R_out || N <= 0) return WN_ERR_ARG;
    if (N % 2) return WN_ERR_BAD_FILTER;
    if (N > 64) return WN_ERR_UNSUPPORTED;
    *R_out = wn::reg_term(a, N, do_wavelet != 0);
    return WN_OK;
}

int wn_reg_term_deriv(const double* a, int N, int do_wavelet, double* grad) {
    if (!a || !grad || N <= 0) return WN_ERR_ARG;
    if (N % 2) return WN_ERR_BAD_FILTER;
    if (N > 64) return WN_ERR_UNSUPPORTED;
    wn::reg_term_deriv(a, N, do_wavelet != 0, grad);
    return WN_OK;
}

// ---- cost over a grid of filters ------------------------------------------------------------------------

int wn_sparse_cost_grid(wn_context* ctx, const double* X, int B, int R, int C, const double* filters, int F, int N,
                        double* S) {
    wn::make_current(ctx);
    int lgR, lgC, sc;
    int rc = check_shape(ctx, B, R, C, false, &lgR, &lgC, &sc);
    if (rc) return rc;
    if (F < 0) return fail(ctx, WN_ERR_ARG, "negative filter count");
    if (N <= 0) return fail(ctx, WN_ERR_BAD_FILTER, "Input filter is empty.");
    if (N % 2) return fail(ctx, WN_ERR_BAD_FILTER, "Input filter length is not a multiple of 2.");
    const int NP = compiled_taps(N);
    if (NP < 0) return fail(ctx, WN_ERR_UNSUPPORTED, "filter length " + std::to_string(N) + " exceeds " + std::to_string(WN_MAX_TAPS));
    if (sc < 0) return fail(ctx, WN_ERR_UNSUPPORTED, "cost grid: examples beyond " + std::to_string(WN_MAX_FUSED) + " samples are not supported");
    if (B == 0 || F == 0) return WN_OK;
    if (!X || !filters || !S) return fail(ctx, WN_ERR_ARG, "null pointer");
    const int chunk = F < 4096 ? F : 4096;
    if (ctx->grid_filters_cap < chunk) {
        WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->grid_filters) (void)hipFree(ctx->grid_filters);
        ctx->grid_filters = nullptr;
        ctx->grid_filters_cap = 0;
        WN_HIP(ctx, hipMalloc((void**)&ctx->grid_filters, (size_t)chunk * sizeof(wn::FilterDev)));
        ctx->grid_filters_cap = chunk;
    }
    const void* w64 = (lgR == 6 && lgC == 6 && ctx->wave64) ? wn_kernel_wave64_transform(4, NP) : nullptr;
    const size_t n = (size_t)R * C;
    if (!w64 && ctx->grid_scratch_count < (size_t)B * n) {
        WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->grid_scratch) (void)hipFree(ctx->grid_scratch);
        ctx->grid_scratch = nullptr;
        ctx->grid_scratch_count = 0;
        WN_HIP(ctx, hipMalloc((void**)&ctx->grid_scratch, (size_t)B * n * sizeof(double)));
        ctx->grid_scratch_count = (size_t)B * n;
    }
    std::vector<wn::FilterDev> table((size_t)chunk);
    const int d = (NP - N) / 2;
    for (int f0 = 0; f0 < F; f0 += chunk) {
        const int Fc = (F - f0 < chunk) ? F - f0 : chunk;
        memset(table.data(), 0, (size_t)Fc * sizeof(wn::FilterDev));
        for (int f = 0; f < Fc; f++) {
            const double* a = filters + (size_t)(f0 + f) * N;
            wn::FilterDev& h = table[(size_t)f];
            for (int j = 0; j < N; j++) {
                h.a_true[j] = a[j];
                h.a[j + d] = a[j];
                h.b[j + d] = wn::parity_sign(j) * a[N - 1 - j];
                h.vL[j + d] = 1.0;
                h.sH[j + d] = wn::parity_sign(N - 1 - j);
            }
            h.N = N; h.NP = NP; h.d = d;
        }
        // the table is pageable host memory and is reused by the next chunk: wait for the kernels that read
        // the previous one, copy synchronously
        WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
        WN_HIP(ctx, hipMemcpy(ctx->grid_filters, table.data(), (size_t)Fc * sizeof(wn::FilterDev), hipMemcpyHostToDevice));
        wn::FusedArgs A;
        memset(&A, 0, sizeof(A));
        A.B = B; A.lgR = lgR; A.lgC = lgC; A.X = X;
        if (w64) {
            A.filt = ctx->grid_filters; A.F = Fc; A.S = S + (size_t)f0 * B;
            const long long waves = (long long)Fc * B;
            rc = launch(ctx, w64, 256, (size_t)4 * wn::W64_TILE * sizeof(double), (int)std::min<long long>((waves + 3) / 4, 1 << 30), A, nullptr);
            if (rc) return rc;
        } else {
            const SizeClass& Sc = kSizeClasses[sc];
            for (int f = 0; f < Fc; f++) {
                wn::FusedArgs Af = A;
                Af.filt = ctx->grid_filters + f; Af.Y = ctx->grid_scratch;
                rc = launch(ctx, fused_kernel(NP, KIND_FORWARD, sc), Sc.T, (size_t)4 * n * sizeof(double), B, Af, nullptr);
                if (rc) return rc;
                wn::FusedArgs As;
                memset(&As, 0, sizeof(As));
                As.B = B; As.lgR = lgR + lgC; As.lgC = 0; As.Yin = ctx->grid_scratch; As.S = S + (size_t)(f0 + f) * B;
                const size_t shmem = ((size_t)Sc.NS + Sc.T / 64 + 2) * sizeof(double) + 16;
                rc = launch(ctx, wn_kernel_sparse(sc), Sc.T, shmem, B, As, nullptr);
                if (rc) return rc;
            }
        }
    }
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `table` goes out of scope
    return WN_OK;
}

// ---- training ---------------------------------------------------------------------------------------

// The LDS-resident and one-wave-per-example training kernels plus the batch reduction: per-workgroup (per-wave)
// sums go to `partial` (room for max_slots slots), the N + 2 sums to `sums`.
static int train_fused(wn_context* ctx, const double* X, int B, int lgR, int lgC, int sc, double* grad, double* Sout,
                       double* partial, int max_slots, double* sums, const wn::StepTail* tail = nullptr) {
    const int R = 1 << lgR, C = 1 << lgC;
    const SizeClass& S = kSizeClasses[sc < 0 ? 0 : sc];
    wn::FusedArgs A;
    memset(&A, 0, sizeof(A));
    A.B = B; A.lgR = lgR; A.lgC = lgC; A.filt = ctx->dfilt;
    A.X = X; A.grad = grad; A.S = Sout; A.partial = partial;
    const void* wave64 = (lgR == 6 && lgC == 6 && ctx->wave64) ? wn_kernel_wave64(ctx->NP) : nullptr;
    const bool wave1d = !wave64 && lgR == 10 && lgC == 0 && ctx->wave64;
    // the last workgroup of k_train64 and of k_train1d1024 ends the step itself (wn_step.h); the kernels of many
    // small workgroups leave that to k_finish
    const bool in_kernel_tail = tail && B > 0 && (wave64 || wave1d);
    if (in_kernel_tail) A.tail = *tail;
    int grid = 0, rc = WN_OK;
    if (B > 0) {
        const bool prof = ctx->prof_used < (int)ctx->prof_begin.size();
        if (prof) WN_HIP(ctx, hipEventRecord(ctx->prof_begin[ctx->prof_used], ctx->stream));
        if (wave64) {
            // one wave per example, four examples per workgroup; one partial-sum slot per workgroup
            rc = launch(ctx, wave64, 256, (size_t)4 * wn::W64_TILE * sizeof(double), (B + 3) / 4, A, &grid, max_slots);
        } else if ((lgR == 8 || lgR == 9) && lgC == 0 && ctx->NP <= 8 && ctx->wave64) {
            // 1D signals of 256 and 512 samples, up to 8 taps: one wave per example, cascade in registers
            rc = launch(ctx, wn_kernel_wave1d_line(ctx->NP, 1 << (lgR - 6)), 64, (512 + 8) * sizeof(double), B, A, &grid, max_slots);
        } else if (wave1d) {
            // 1D signals of 1024 samples: one wave per example, four examples per workgroup (an LDS slice per wave);
            // one partial-sum slot per workgroup
            rc = launch(ctx, wn_kernel_wave1d(ctx->NP), 256, (size_t)4 * (1024 + 64 + 256 + 8 + (ctx->NP == 16 ? 15 * 64 : 0)) * sizeof(double),
                        (B + 3) / 4, A, &grid, max_slots);
        } else {
            // X, row-stage coefficients (2D only), pyramid, coefficients / sort slots, reduction scratch
            const size_t shmem = ((size_t)(lgC > 0 ? 3 : 2) * R * C + S.NS + (size_t)(S.T / 64) * ctx->NP + 2) * sizeof(double) + 16;
            rc = launch(ctx, fused_kernel(ctx->NP, KIND_TRAIN, sc), S.T, shmem, B, A, &grid, max_slots);
        }
        if (rc) return rc;
        if (prof) {
            WN_HIP(ctx, hipEventRecord(ctx->prof_end[ctx->prof_used], ctx->stream));
            ctx->prof_used++;
        }
    }
    if (in_kernel_tail) return WN_OK;
    if (tail) {
        hipLaunchKernelGGL(k_finish, dim3(1), dim3(1024), 0, ctx->stream, partial, grid, *tail);
        WN_HIP(ctx, hipGetLastError());
        return WN_OK;
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3(ctx->N + 2), dim3(64), 0, ctx->stream, partial, grid, ctx->N, sums);
    WN_HIP(ctx, hipGetLastError());
    return WN_OK;
}

int wn_train_batch(wn_context* ctx, const double* X, int B, int R, int C, double* grad, double* Sout) {
    wn::make_current(ctx);
    int lgR, lgC, sc;
    int rc = check_shape(ctx, B, R, C, true, &lgR, &lgC, &sc);
    if (rc) return rc;
    if (B > 0 && !X) return fail(ctx, WN_ERR_ARG, "null input");
    if (sc < 0 && B > 0) {
        // streamed path: the profiling record spans all of its kernels
        const bool prof = ctx->prof_used < (int)ctx->prof_begin.size();
        if (prof) WN_HIP(ctx, hipEventRecord(ctx->prof_begin[ctx->prof_used], ctx->stream));
        rc = wn::large_train(ctx, X, B, lgR, lgC, grad, Sout);
        if (rc) return rc;
        if (prof) {
            WN_HIP(ctx, hipEventRecord(ctx->prof_end[ctx->prof_used], ctx->stream));
            ctx->prof_used++;
        }
        return WN_OK;
    }
    // the sums are added up where wn_train_step adds them (mode 1: no update) -- for 64x64 images inside the training
    // kernel, which saves the data-parallel loop (this call, all-reduce, wn_apply_update) a launch per batch
    wn::StepTail T{};
    T.ticket = ctx->ticket;
    T.sums = ctx->sums;
    T.filt = ctx->dfilt;
    T.mode = 1;
    return train_fused(ctx, X, B, lgR, lgC, sc, grad, Sout, ctx->partial, ctx->partial_blocks, ctx->sums, B > 0 ? &T : nullptr);
}

int wn_train_batch_static(wn_context* ctx, const double* X, int B, int R, int C, double* sums_dev) {
    wn::make_current(ctx);
    int lgR, lgC, sc;
    int rc = check_shape(ctx, B, R, C, true, &lgR, &lgC, &sc);
    if (rc) return rc;
    if (!sums_dev) return fail(ctx, WN_ERR_ARG, "null output");
    if (B > 0 && !X) return fail(ctx, WN_ERR_ARG, "null input");
    if (sc < 0)
        return fail(ctx, WN_ERR_UNSUPPORTED,
                    "wn_train_batch_static: examples of " + std::to_string((long long)R * C) + " samples exceed the " +
                        std::to_string(WN_MAX_FUSED) + " it takes (larger ones need a workspace that grows with the batch: use wn_train_batch)");
    std::lock_guard<std::mutex> lock(scratch_mutex);
    double* part = scratch_ptr[ctx->device];
    if (!part) return fail(ctx, WN_ERR_STATE, "the static scratch array was not located on this device");
    wn_context* last = scratch_last[ctx->device];
    if (last && last != ctx) WN_HIP(ctx, hipStreamWaitEvent(ctx->stream, last->static_done, 0));
    rc = train_fused(ctx, X, B, lgR, lgC, sc, nullptr, nullptr, part, kStaticSlots, sums_dev);
    if (rc == WN_OK && B > 0) {
        WN_HIP(ctx, hipEventRecord(ctx->static_done, ctx->stream));
        scratch_last[ctx->device] = ctx;
    }
    return rc;
}

int wn_sums(wn_context* ctx, double** sums_dev, int* count) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    if (!ctx->has_filter) return fail(ctx, WN_ERR_STATE, "no filter set");
    if (sums_dev) *sums_dev = ctx->sums;
    if (count) *count = ctx->N + 2;
    return WN_OK;
}

int wn_set_sums_buffer(wn_context* ctx, double* sums_dev) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->sums = sums_dev ? sums_dev : ctx->own_sums;
    return WN_OK;
}

int wn_profile_begin(wn_context* ctx, int max_records) {
    wn::make_current(ctx);
    if (!ctx || max_records < 0) return WN_ERR_ARG;
    WN_HIP(ctx, hipSetDevice(ctx->device));
    while ((int)ctx->prof_begin.size() < max_records) {
        hipEvent_t a, b;
        WN_HIP(ctx, hipEventCreate(&a));
        WN_HIP(ctx, hipEventCreate(&b));
        ctx->prof_begin.push_back(a);
        ctx->prof_end.push_back(b);
    }
    while ((int)ctx->prof_begin.size() > max_records) {
        (void)hipEventDestroy(ctx->prof_begin.back());
        (void)hipEventDestroy(ctx->prof_end.back());
        ctx->prof_begin.pop_back();
        ctx->prof_end.pop_back();
    }
    ctx->prof_used = 0;
    return WN_OK;
}

int wn_profile_collect(wn_context* ctx, float* ms_host, int capacity, int* count) {
    wn::make_current(ctx);
    if (!ctx || !count || capacity < 0) return WN_ERR_ARG;
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int n = ctx->prof_used < capacity ? ctx->prof_used : capacity;
    for (int i = 0; i < n; i++) {
        float ms = 0.f;
        WN_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_begin[i], ctx->prof_end[i]));
        if (ms_host) ms_host[i] = ms;
    }
    *count = n;
    ctx->prof_used = 0;
    return WN_OK;
}

int wn_get_sums(wn_context* ctx, double* sums_host, int* count) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    if (!ctx->has_filter) return fail(ctx, WN_ERR_STATE, "no filter set");
    if (count) *count = ctx->N + 2;
    if (sums_host) return wn_memcpy_d2h(ctx, sums_host, ctx->sums, (size_t)(ctx->N + 2) * sizeof(double));
    return WN_OK;
}

static int update_params(wn_context* ctx, double batch_total, double lambda_sum, double alpha, double inertia_eff,
                         int do_wavelet, wn::UpdateParams* P) {
    if (!ctx->has_filter) return fail(ctx, WN_ERR_STATE, "no filter set");
    if (!(batch_total >= 1)) return fail(ctx, WN_ERR_ARG, "batch_total must be >= 1");
    P->Bf = (double)(float)batch_total;  // the reference divides by float(size)
    P->lambda_sum = lambda_sum;
    P->alpha = alpha;
    P->inertia_eff = inertia_eff;
    P->wavelet = do_wavelet;
    P->cap = kLogCap;
    P->logF = ctx->log_filters;
    P->logC = ctx->log_costs;
    return WN_OK;
}

int wn_apply_update(wn_context* ctx, double batch_total, double lambda_sum, double alpha, double inertia_eff,
                    int do_wavelet) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    wn::UpdateParams P;
    const int rc = update_params(ctx, batch_total, lambda_sum, alpha, inertia_eff, do_wavelet, &P);
    if (rc) return rc;
    hipLaunchKernelGGL(k_update, dim3(1), dim3(64), 0, ctx->stream, ctx->dfilt, ctx->sums, P);
    WN_HIP(ctx, hipGetLastError());
    return WN_OK;
}

int wn_train_step(wn_context* ctx, const double* X, int B, int R, int C, double batch_total, double lambda_sum,
                  double alpha, double inertia_eff, int do_wavelet) {
    wn::make_current(ctx);
    int lgR, lgC, sc;
    int rc = check_shape(ctx, B, R, C, true, &lgR, &lgC, &sc);
    if (rc) return rc;
    if (B > 0 && !X) return fail(ctx, WN_ERR_ARG, "null input");
    wn::UpdateParams P;
    rc = update_params(ctx, batch_total, lambda_sum, alpha, inertia_eff, do_wavelet, &P);
    if (rc) return rc;
    const bool exchange = ctx->comm && ctx->nranks > 1;
    if (sc >= 0) {
        // one launch: the training kernel's last workgroup reduces the partial sums and, without a
        // communicator, applies the update
        wn::StepTail T;
        T.ticket = ctx->ticket;
        T.sums = ctx->sums;
        T.filt = ctx->dfilt;
        T.mode = exchange ? 1 : 2;
        T.P = P;
        rc = train_fused(ctx, X, B, lgR, lgC, sc, nullptr, nullptr, ctx->partial, ctx->partial_blocks, ctx->sums, &T);
        if (rc || !exchange) return rc;
    } else {
        rc = wn_train_batch(ctx, X, B, R, C, nullptr, nullptr);
        if (rc) return rc;
    }
    if (exchange && (rc = wn_allreduce_sums(ctx))) return rc;
    hipLaunchKernelGGL(k_update, dim3(1), dim3(64), 0, ctx->stream, ctx->dfilt, ctx->sums, P);
    WN_HIP(ctx, hipGetLastError());
    return WN_OK;
}

// steps and the non-finite flag sit side by side in FilterDev
static int read_update_state(wn_context* ctx, int* steps, int* nonfinite) {
    int two[2] = {0, 0};
    static_assert(offsetof(wn::FilterDev, nonfinite) == offsetof(wn::FilterDev, steps) + sizeof(int), "layout");
    const int rc = wn_memcpy_d2h(ctx, two, &ctx->dfilt->steps, sizeof(two));
    *steps = two[0];
    *nonfinite = two[1];
    return rc;
}

int wn_num_updates(wn_context* ctx, int* steps) {
    wn::make_current(ctx);
    if (!ctx || !steps) return WN_ERR_ARG;
    int bad = 0;
    const int rc = read_update_state(ctx, steps, &bad);
    if (rc) return rc;
    if (bad)
        return fail(ctx, WN_ERR_NONFINITE, "a batch gave non-finite sums and was not applied (most likely a diverging "
                                           "solution: lower alpha or lambda); no update is applied until wn_set_filter or wn_clear_logs");
    return WN_OK;
}

// Gathers the update state on the device (k_pack_updates) and brings it over in ONE copy into the context's pinned
// buffer; returns the device's update count and flag.  rows: log entries wanted from `first` on.
static int fetch_updates(wn_context* ctx, int first, int rows, int* steps, int* nonfinite) {
    constexpr size_t ROW = wn::MAXTAPS + 1;
    const size_t need = 2 + wn::MAXTAPS + (size_t)rows * ROW;
    if (need > ctx->pack_doubles) {
        if (ctx->pack_dev) (void)hipFree(ctx->pack_dev);
        if (ctx->pack_host) (void)hipHostFree(ctx->pack_host);
        ctx->pack_dev = ctx->pack_host = nullptr;
        ctx->pack_doubles = 0;
        const size_t cap = std::max<size_t>(need, 2 + wn::MAXTAPS + 1024 * ROW);
        WN_HIP(ctx, hipMalloc((void**)&ctx->pack_dev, cap * sizeof(double)));
        WN_HIP(ctx, hipHostMalloc((void**)&ctx->pack_host, cap * sizeof(double), hipHostMallocDefault));
        ctx->pack_doubles = cap;
    }
    const int blocks = std::max(1, std::min(64, (int)((size_t)rows * ROW / 256)));
    hipLaunchKernelGGL(k_pack_updates, dim3(blocks), dim3(256), 0, ctx->stream, ctx->dfilt, ctx->log_filters, ctx->log_costs, first, rows,
                       kLogCap, ctx->pack_dev);
    WN_HIP(ctx, hipGetLastError());
    WN_HIP(ctx, hipMemcpyAsync(ctx->pack_host, ctx->pack_dev, need * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *steps = (int)ctx->pack_host[0];
    *nonfinite = (int)ctx->pack_host[1];
    return WN_OK;
}

static void unpack_rows(const wn_context* ctx, int count, double* filters, double* costs) {
    constexpr size_t ROW = wn::MAXTAPS + 1;
    const double* rows = ctx->pack_host + 2 + wn::MAXTAPS;
    const int N = ctx->N;
    for (int i = 0; i < count; i++) {
        if (filters) memcpy(filters + (size_t)i * N, rows + (size_t)i * ROW, (size_t)N * sizeof(double));
        if (costs) costs[i] = rows[(size_t)i * ROW + wn::MAXTAPS];
    }
}

int wn_get_logs(wn_context* ctx, int first, int count, double* filters, double* costs) {
    wn::make_current(ctx);
    if (!ctx || first < 0 || count < 0) return WN_ERR_ARG;
    if (!ctx->has_filter) return fail(ctx, WN_ERR_STATE, "no filter set");
    int steps = 0, bad = 0;
    const int rc = fetch_updates(ctx, first, count, &steps, &bad);
    if (rc) return rc;
    if (first + count > steps || steps - first > kLogCap)
        return fail(ctx, WN_ERR_ARG, "log range not available (" + std::to_string(steps) + " updates, ring of " +
                                         std::to_string(kLogCap) + ")");
    unpack_rows(ctx, count, filters, costs);
    return WN_OK;
}

int wn_get_updates(wn_context* ctx, int first, int max_count, int* steps_out, double* filters, double* costs, double* momentum) {
    wn::make_current(ctx);
    if (!ctx || first < 0 || max_count < 0 || !steps_out) return WN_ERR_ARG;
    if (!ctx->has_filter) return fail(ctx, WN_ERR_STATE, "no filter set");
    int steps = 0, bad = 0;
    const int rc = fetch_updates(ctx, first, max_count, &steps, &bad);
    if (rc) return rc;
    *steps_out = steps;
    if (steps < first || steps - first > kLogCap)
        return fail(ctx, WN_ERR_ARG, "log range not available (" + std::to_string(steps) + " updates, ring of " +
                                         std::to_string(kLogCap) + ")");
    unpack_rows(ctx, std::min(max_count, steps - first), filters, costs);
    if (momentum) memcpy(momentum, ctx->pack_host + 2, (size_t)ctx->N * sizeof(double));
    if (bad)
        return fail(ctx, WN_ERR_NONFINITE, "a batch gave non-finite sums and was not applied (most likely a diverging "
                                           "solution: lower alpha or lambda); no update is applied until wn_set_filter or wn_clear_logs");
    return WN_OK;
}

int wn_clear_logs(wn_context* ctx) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    hipLaunchKernelGGL(k_reset_steps, dim3(1), dim3(64), 0, ctx->stream, ctx->dfilt);
    WN_HIP(ctx, hipGetLastError());
    return WN_OK;
}

// ---- RCCL ---------------------------------------------------------------------------------------------

int wn_comm_unique_id(void* id128) {
    if (!id128) return WN_ERR_ARG;
    void* lib = open_rccl();
    if (!lib) { wn::global_error() = std::string("dlopen librccl: ") + dlerror(); return WN_ERR_COMM; }
    auto f = (fn_ncclGetUniqueId)dlsym(lib, "ncclGetUniqueId");
    if (!f) { wn::global_error() = "ncclGetUniqueId not found"; return WN_ERR_COMM; }
    NcclId id;
    const int r = f(&id);
    if (r != 0) { wn::global_error() = "ncclGetUniqueId failed: " + std::to_string(r); return WN_ERR_COMM; }
    memcpy(id128, &id, sizeof(id));
    return WN_OK;
}

int wn_comm_init(wn_context* ctx, const void* id128, int nranks, int rank) {
    wn::make_current(ctx);
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return WN_ERR_ARG;
    if (ctx->comm) return fail(ctx, WN_ERR_STATE, "communicator already initialised");
    void* lib = open_rccl();
    if (!lib) return fail(ctx, WN_ERR_COMM, std::string("dlopen librccl: ") + dlerror());
    auto f = (fn_ncclCommInitRank)dlsym(lib, "ncclCommInitRank");
    if (!f) return fail(ctx, WN_ERR_COMM, "ncclCommInitRank not found");
    WN_HIP(ctx, hipSetDevice(ctx->device));
    NcclId id;
    memcpy(&id, id128, sizeof(id));
    void* comm = nullptr;
    const int r = f(&comm, nranks, id, rank);
    if (r != 0) return fail(ctx, WN_ERR_COMM, "ncclCommInitRank failed: " + std::to_string(r));
    ctx->nccl_lib = lib;
    ctx->comm = comm;
    ctx->nranks = nranks;
    ctx->fn_allreduce = dlsym(lib, "ncclAllReduce");
    if (!ctx->fn_allreduce) return fail(ctx, WN_ERR_COMM, "ncclAllReduce not found");
    return WN_OK;
}

// One process, one host thread, several GPUs: all communicators at once (ncclCommInitAll) and the all-reduce
// of every context inside one group call, so that no call waits for a rank the same thread has yet to serve.
typedef int (*fn_ncclCommInitAll)(void**, int, const int*);
typedef int (*fn_ncclGroup)(void);

int wn_comm_init_all(wn_context** ctxs, int n) {
    if (!ctxs || n < 1 || n > 64) return WN_ERR_ARG;
    for (int r = 0; r < n; r++) {
        if (!ctxs[r]) return WN_ERR_ARG;
        if (ctxs[r]->comm) return fail(ctxs[r], WN_ERR_STATE, "communicator already initialised");
        for (int q = 0; q < r; q++) {
            if (ctxs[q]->device == ctxs[r]->device)
                return fail(ctxs[r], WN_ERR_ARG, "RCCL needs every context on a GPU of its own (device " + std::to_string(ctxs[r]->device) + " appears twice)");
        }
    }
    void* lib = open_rccl();
    if (!lib) return fail(ctxs[0], WN_ERR_COMM, std::string("dlopen librccl: ") + dlerror());
    auto f = (fn_ncclCommInitAll)dlsym(lib, "ncclCommInitAll");
    if (!f) return fail(ctxs[0], WN_ERR_COMM, "ncclCommInitAll not found");
    std::vector<void*> comms((size_t)n, nullptr);
    std::vector<int> devs((size_t)n);
    for (int r = 0; r < n; r++) devs[(size_t)r] = ctxs[r]->device;
    const int rc = f(comms.data(), n, devs.data());
    if (rc != 0) return fail(ctxs[0], WN_ERR_COMM, "ncclCommInitAll failed: " + std::to_string(rc));
    void* ar = dlsym(lib, "ncclAllReduce");
    void* gs = dlsym(lib, "ncclGroupStart");
    void* ge = dlsym(lib, "ncclGroupEnd");
    if (!ar || !gs || !ge) return fail(ctxs[0], WN_ERR_COMM, "ncclAllReduce / ncclGroupStart / ncclGroupEnd not found");
    for (int r = 0; r < n; r++) {
        ctxs[r]->nccl_lib = lib;
        ctxs[r]->comm = comms[(size_t)r];
        ctxs[r]->nranks = n;
        ctxs[r]->fn_allreduce = ar;
        ctxs[r]->fn_group_start = gs;
        ctxs[r]->fn_group_end = ge;
    }
    return WN_OK;
}

int wn_allreduce_sums_all(wn_context** ctxs, int n) {
    if (!ctxs || n < 1) return WN_ERR_ARG;
    for (int r = 0; r < n; r++) {
        if (!ctxs[r]) return WN_ERR_ARG;
        if (!ctxs[r]->has_filter) return fail(ctxs[r], WN_ERR_STATE, "no filter set");
        if (!ctxs[r]->comm || ctxs[r]->nranks != n) return fail(ctxs[r], WN_ERR_STATE, "no communicator over these contexts: call wn_comm_init_all first");
    }
    // (entry points looked up once, in wn_comm_init_all: this call sits on the serial path of every batch)
    auto ar = (fn_ncclAllReduce)ctxs[0]->fn_allreduce;
    auto gs = (fn_ncclGroup)ctxs[0]->fn_group_start;
    auto ge = (fn_ncclGroup)ctxs[0]->fn_group_end;
    if (!ar || !gs || !ge) return fail(ctxs[0], WN_ERR_STATE, "no grouped communicator over these contexts: call wn_comm_init_all first");
    int rc = gs();
    for (int r = 0; r < n && rc == 0; r++) {
        wn_context* c = ctxs[r];
        rc = ar(c->sums, c->sums, (size_t)(c->N + 2), kNcclDouble, kNcclSum, c->comm, c->stream);
    }
    const int rc2 = ge();
    if (rc != 0 || rc2 != 0) return fail(ctxs[0], WN_ERR_COMM, "grouped ncclAllReduce failed: " + std::to_string(rc ? rc : rc2));
    return WN_OK;
}

int wn_comm_destroy(wn_context* ctx) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    if (ctx->comm && ctx->nccl_lib) {
        auto f = (fn_ncclCommDestroy)dlsym(ctx->nccl_lib, "ncclCommDestroy");
        if (f) f(ctx->comm);
    }
    ctx->comm = nullptr;
    ctx->nranks = 1;
    return WN_OK;
}

int wn_allreduce_sums(wn_context* ctx) {
    wn::make_current(ctx);
    if (!ctx) return WN_ERR_ARG;
    if (!ctx->has_filter) return fail(ctx, WN_ERR_STATE, "no filter set");
    if (!ctx->comm) return fail(ctx, WN_ERR_STATE, "no communicator: call wn_comm_init first");
    auto f = (fn_ncclAllReduce)ctx->fn_allreduce;     // looked up when the communicator was made
    if (!f) return fail(ctx, WN_ERR_COMM, "ncclAllReduce not found");
    const int r = f(ctx->sums, ctx->sums, (size_t)(ctx->N + 2), kNcclDouble, kNcclSum, ctx->comm, ctx->stream);
    if (r != 0) return fail(ctx, WN_ERR_COMM, "ncclAllReduce failed: " + std::to_string(r));
    return WN_OK;
}

}  // extern "C"
