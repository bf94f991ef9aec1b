// Shared between the translation units of libwavenet_hip.so: the context behind the opaque
// wn_context handle, error plumbing, and the entry points of the streamed (large-shape) path.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

#include "wavenet_hip.h"
#include "wn_device.h"

struct wn_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int numCU = 0;
    wn::FilterDev* dfilt = nullptr;
    bool has_filter = false;
    int N = 0, NP = 0;
    double* partial = nullptr;
    int partial_blocks = 0;
    void* gen_tables = nullptr;               // device copies of the generators' tables (Poisson, energy, exp) ...
    void* gen_tables_host = nullptr;          // ... and what was last uploaded, so that an unchanged table is not sent again
    wn::FilterDev* grid_filters = nullptr;   // table of filters for wn_sparse_cost_grid
    int grid_filters_cap = 0;
    double* grid_scratch = nullptr;           // coefficients of one batch (shapes without a fused cost kernel)
    size_t grid_scratch_count = 0;
    bool bucket_rank = true;  // streamed path: ranks by histogram + per-bucket LDS sort (WAVENET_HIP_RANK=radix: device-wide radix sort)
    bool rank_configured = false;   // the ranking kernels' LDS limits have been raised on this context's device
    int row_line_taps = 16;   // streamed path: the row passes run on the one-wave-per-line kernels up to this many taps (WAVENET_HIP_ROW_LINE_TAPS)
    bool fused_hist = true;   // streamed path, 2D on lines: the column kernel builds the ranking's histograms (WAVENET_HIP_FUSED_HIST=0: kr_hist)
    bool rank_pairs_cols = true;  // streamed path, 2D on lines: the column backward kernel takes (home index, rank) pairs region by region (WAVENET_HIP_RANK_PAIRS=0: ranks at home positions)
    int rank_pairs_taps = 8;      // ... up to this many taps (WAVENET_HIP_RANK_PAIRS_TAPS)
    int rank_regions_from = 19;   // streamed path, errors wanted: from 2^this coefficients per example the ranks go home region by region (WAVENET_HIP_RANK_REGIONS_FROM)
    int rank_regions_ib = 14;     // log2 of the positions per region then (WAVENET_HIP_RANK_REGIONS_IB, 13 .. 15)
    bool regions_configured = false;
    bool rank_packed = true;  // streamed path: 8-byte packed staging keys (WAVENET_HIP_RANK_KEYS=12: magnitudes and indices side by side)
    int rank_sub = 0;         // streamed path: examples per partition + bucket-sort launch pair (WAVENET_HIP_RANK_SUB; 0: the whole chunk)
    int chunk_cap = 0;        // streamed path: examples per chunk at most (WAVENET_HIP_CHUNK; 0: as many as the workspace limit allows)
    bool exact_sort = false;  // streamed path: always the 64-bit sort (WAVENET_HIP_EXACT_SORT=1; for tests)
    bool wave64 = true;  // 64x64 training on the one-wave-per-example kernel (WAVENET_HIP_WAVE64=0 turns it off)
    double* sums = nullptr;
    double* own_sums = nullptr;
    // streamed path: the ranking's give-up flag is fetched on a stream of its own as soon as the ranking kernels are done
    // (an event on the main stream), while the main stream goes on with the backward pass
    hipStream_t side_stream = nullptr;
    hipEvent_t rank_event = nullptr;
    hipEvent_t static_done = nullptr;   // recorded after each wn_train_batch_static of this context
    unsigned* ticket = nullptr;  // counter of finished workgroups of a training kernel whose last workgroup ends the step
    std::vector<hipEvent_t> prof_begin, prof_end;
    int prof_used = 0;
    double* log_filters = nullptr;
    double* log_costs = nullptr;
    double* pack_dev = nullptr;      // wn_get_updates: the update state gathered on the device ...
    double* pack_host = nullptr;     // ... and its pinned host copy (pack_doubles each)
    size_t pack_doubles = 0;
    void* fn_allreduce = nullptr;    // RCCL entry points, looked up once when the communicator is made
    void* fn_group_start = nullptr;
    void* fn_group_end = nullptr;
    void* nccl_lib = nullptr;
    void* comm = nullptr;
    int nranks = 1;
    std::string err;
    std::map<const void*, size_t> lds_configured;
    std::map<std::pair<const void*, size_t>, int> occupancy;
    // streamed path: one grow-only device workspace
    void* workspace = nullptr;
    size_t workspace_bytes = 0;
};

namespace wn {

// Every entry point that takes a context starts with this: one host thread may drive contexts on several GPUs (the
// host layer's data-parallel Coach does), and allocations, attribute changes and launches go to the current device.
inline void make_current(wn_context* ctx) {
    if (ctx) (void)hipSetDevice(ctx->device);
}

inline std::string& global_error() {
    static thread_local std::string e;  // for the entry points that take no context
    return e;
}

inline int fail(wn_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    global_error() = msg;
    return code;
}

// Streamed path for examples beyond the LDS-resident limit (R*C > WN_MAX_FUSED): every level of the
// cascade is one kernel over the whole batch chunk, intermediates live in a device workspace.
int large_forward(wn_context* ctx, const double* X, int B, int lgR, int lgC, double* Y, double* Prow, double* Y1, double* Pcol);
int large_inverse(wn_context* ctx, const double* Y, int B, int lgR, int lgC, double* X);
int large_backprop(wn_context* ctx, const double* Delta, const double* X, const double* Prow, const double* Y1,
                   const double* Pcol, int B, int lgR, int lgC, double* grad, double* din);
int large_sparse(wn_context* ctx, const double* Y, int B, long long n, double* Delta, double* S);
// Leaves the N + 2 sums in ctx->sums like the fused path.
int large_train(wn_context* ctx, const double* X, int B, int lgR, int lgC, double* grad, double* S);

}  // namespace wn

#define WN_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return wn::fail(ctx, WN_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)
