/*
 * wavenet_hip.h -- C ABI of the MI355X (gfx950) implementation of the Wavenet training path.
 *
 * This is the boundary a maintainer of the reference binds against: every entry point names
 * the reference interface it stands in for (file:line under the reference tree).  The
 * reference has no FFI layer; its boundary is the in-process C++ API of
 * include/Wavenet/{Wavenet,CostFunctions,LowpassOperator,HighpassOperator,MatrixOperator}.h,
 * which wavenet_amd/host/ re-creates on top of this header.
 *
 * Conventions
 *   - every array is fp64; an example is an R x C matrix stored column-major (X(r,c) at
 *     r + c*R, as arma::Mat<double>), examples of a batch are contiguous (X + e*R*C);
 *     R and C are powers of two; a 1D signal of length n is R = n, C = 1
 *     (src/GeneratorBase.cxx:25-31 squeezes {1,n} to {n,1});
 *   - pointers named *_dev are device pointers on the context's GPU, *_host are host pointers;
 *   - all work is enqueued on the context's stream and is asynchronous unless stated;
 *   - every function returns WN_OK or an error code and never throws; wn_last_error() gives
 *     the message of the last failure on that context;
 *   - a context is not re-entrant: one host thread at a time.
 *
 * Activations of a batch are exchanged as three arrays shaped like the input:
 *   Prow  low-pass pyramid of every row:    Prow[r + R*(2^i + k)] = activations[0][r](i,0)(k)
 *   Y1    coefficients after the row stage (the column stage's input)
 *   Pcol  low-pass pyramid of every column: Pcol[(2^i + k) + R*c] = activations[1][c](i,0)(k)
 * for levels i = 0 .. log2(len)-1; index 0 of each line is zero.  Together with X and Y they
 * carry everything in the reference's Activations2D_t (src/Wavenet.cxx:569-580): the
 * high-pass activations are the coefficients themselves (src/Utilities.cxx:24-41).
 */
#ifndef WAVENET_HIP_H
#define WAVENET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    WN_OK = 0,
    WN_ERR_ARG = 1,          /* null pointer, negative count                                   */
    WN_ERR_BAD_SHAPE = 2,    /* R or C not a power of two (src/GeneratorBase.cxx:15-49)        */
    WN_ERR_BAD_FILTER = 3,   /* empty or odd-length filter (src/Wavenet.cxx:11-19)             */
    WN_ERR_UNSUPPORTED = 4,  /* filter longer than WN_MAX_TAPS, example beyond 2^30 samples        */
    WN_ERR_HIP = 5,          /* a HIP runtime call failed                                       */
    WN_ERR_COMM = 6,         /* an RCCL call failed                                             */
    WN_ERR_NONFINITE = 7,    /* NaN/Inf in cost or gradient ("diverging solution",
                                src/Wavenet.cxx:234-241): returned by wn_num_updates            */
    WN_ERR_STATE = 8         /* call order: no filter set, no communicator, ...                 */
};

#define WN_MAX_TAPS 32       /* longest filter the compiled kernels take */
#define WN_MAX_FUSED 4096    /* largest R*C of the LDS-resident kernels; larger examples take the streamed
                                path (one kernel per level, device workspace owned by the context) */

typedef struct wn_context wn_context;

/* ---- context ---------------------------------------------------------------------------- */
/* One context per device and caller thread.  Environment read here: WAVENET_HIP_WAVE64=0 switches the
 * one-wave kernels off -- those that give a 64x64 image, a 1024-sample signal or (in larger images, up to
 * 8 taps) a line of 128 ... 1024 samples to one wave and hold it in registers -- and keeps every shape on
 * the workgroup-per-example and level-by-level kernels (same results to rounding; a comparison hook).
 * WAVENET_HIP_EXACT_SORT=1 makes examples beyond 4096 samples rank with two 64-bit radix sorts instead of
 * one sort of 32-bit keys plus the exact re-ranking of equal keys (same ranks). */
int wn_create(int device, wn_context** ctx);
int wn_destroy(wn_context* ctx);
/* A context starts on a stream of its own.  This switches it to the caller's hipStream_t (e.g. the
 * framework's current stream) so that its work is ordered with the caller's; NULL is HIP's default
 * (null) stream. */
int wn_set_stream(wn_context* ctx, void* hip_stream);
int wn_synchronize(wn_context* ctx);
/* The GPU a context was created on. */
int wn_get_device(wn_context* ctx, int* device);
const char* wn_last_error(const wn_context* ctx);
const char* wn_status_string(int status);

/* Plain device-memory helpers so that a host program needs no HIP headers. */
int wn_malloc(wn_context* ctx, void** dev_ptr, size_t bytes);
int wn_free(wn_context* ctx, void* dev_ptr);
int wn_memcpy_h2d(wn_context* ctx, void* dst_dev, const void* src_host, size_t bytes); /* synchronous */
int wn_memcpy_d2h(wn_context* ctx, void* dst_host, const void* src_dev, size_t bytes); /* synchronous */

/* ---- filter operators -------------------------------------------------------------------- */
/*
 * Wavenet::setFilter (src/Wavenet.cxx:8-36) together with everything it invalidates:
 * LowpassOperator(filter, i) / HighpassOperator(filter, i) for every level
 * (Wavenet::cacheOperators_, src/Wavenet.cxx:767-786) and the weight matrices of
 * Wavenet::cacheWeights_ (src/Wavenet.cxx:800-839).  No matrix is built: the N taps and the
 * derived high-pass taps b_i = (-1)^i a_{N-1-i} (src/HighpassOperator.cxx:16-18) are stored on
 * the device for the kernels.  Resets the momentum when N changes, keeps it otherwise; takes the
 * non-finite flag (wn_apply_update) down.
 */
int wn_set_filter(wn_context* ctx, const double* a_host, int N);
int wn_get_filter(wn_context* ctx, double* a_host, int* N);        /* synchronises */
int wn_set_momentum(wn_context* ctx, const double* p_host, int N); /* Wavenet::setMomentum, Wavenet.h:174-178 */
int wn_get_momentum(wn_context* ctx, double* p_host, int* N);      /* synchronises */

/*
 * Dense 2^size x 2^(size+1) column-major matrices, host side, for callers that still want the
 * reference's arma::Mat view: LowpassOperator(filter,size) (include/Wavenet/LowpassOperator.h:44-45)
 * and HighpassOperator(filter,size) (include/Wavenet/HighpassOperator.h:48-49), i.e.
 * MatrixOperator::construct (src/MatrixOperator.cxx:5-113).  Never used by the kernels.
 */
int wn_lowpass_operator(const double* a_host, int N, unsigned size, double* M_host);
int wn_highpass_operator(const double* a_host, int N, unsigned size, double* M_host);

/* ---- transforms -------------------------------------------------------------------------- */
/*
 * Wavenet::forward_(Col) / forward_(Mat) (src/Wavenet.cxx:404-445, 562-622) followed by
 * coeffsFromActivations (src/Utilities.cxx:24-58) on a batch.  Y_dev receives the wavelet
 * coefficients; Prow_dev, Y1_dev, Pcol_dev receive the activations and may each be NULL.
 */
int wn_forward(wn_context* ctx, const double* X_dev, int B, int R, int C,
               double* Y_dev, double* Prow_dev, double* Y1_dev, double* Pcol_dev);

/* Wavenet::inverse_(Col) / inverse_(Mat) (src/Wavenet.cxx:447-472, 624-663). */
int wn_inverse(wn_context* ctx, const double* Y_dev, int B, int R, int C, double* X_dev);

/*
 * Wavenet::backpropagate_(Col, Activations1D_t) / backpropagate_(Mat, Activations2D_t)
 * (src/Wavenet.cxx:474-556, 665-711) with frobeniusProduct / outerProduct
 * (include/Wavenet/Wavenet.h:870-886) on a batch.  grad_dev[e*N + k] receives the filter
 * gradient of example e.  din_dev (may be NULL) receives the errors after the column stage,
 * which for a 1D signal are the reference's "errors on the input" (output.at(1)).
 */
int wn_backprop(wn_context* ctx, const double* Delta_dev, const double* X_dev,
                const double* Prow_dev, const double* Y1_dev, const double* Pcol_dev,
                int B, int R, int C, double* grad_dev, double* din_dev);

/* ---- cost functions ---------------------------------------------------------------------- */
/* SparseTerm(Col/Mat) (src/CostFunctions.cxx:6-46): S_dev[e] = 1 - Gini of example e's n values. */
int wn_sparse_term(wn_context* ctx, const double* Y_dev, int B, int n, double* S_dev);
/* SparseTermDeriv(Col/Mat) (src/CostFunctions.cxx:49-117); S_dev may be NULL. */
int wn_sparse_term_deriv(wn_context* ctx, const double* Y_dev, int B, int n,
                         double* Delta_dev, double* S_dev);
/* RegTerm / RegTermDeriv (src/CostFunctions.cxx:119-257, 259-415), host arithmetic on N taps. */
int wn_reg_term(const double* a_host, int N, int do_wavelet, double* R_out);
int wn_reg_term_deriv(const double* a_host, int N, int do_wavelet, double* grad_host);

/* ---- training ---------------------------------------------------------------------------- */
/*
 * The example-dependent part of Wavenet::train (src/Wavenet.cxx:196-212 and the SparseTerm of
 * :227) for B examples with the current filter: forward, SparseTermDeriv, backpropagate.
 * Leaves N + 2 doubles in the context's sum buffer (see wn_sums): the sum over the batch of the
 * sparsity gradients, the sum of SparseTerm, and the number of non-finite values met.
 * grad_dev ([B][N]) and S_dev ([B]) receive per-example values and may be NULL.
 */
int wn_train_batch(wn_context* ctx, const double* X_dev, int B, int R, int C,
                   double* grad_dev, double* S_dev);

/*
 * wn_train_batch without any allocation.  It has no counterpart in the reference; it is for callers that
 * cannot call hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid it.
 * Same kernels as wn_train_batch, but the per-workgroup partial sums -- the one scratch buffer of that
 * path, which the context otherwise holds in memory it allocated -- go to a fixed array inside the library's
 * code object (WN_STATIC_SLOTS slots of WN_MAX_TAPS + 2 doubles, 2.1 MiB, held by every GPU the library is
 * used on until the process ends), and the N + 2 sums are written to sums_dev, the caller's device memory
 * (not to the context's sum buffer).  No per-example outputs.  Examples of up to WN_MAX_FUSED samples, any
 * B; a larger example is refused with WN_ERR_UNSUPPORTED (its path needs a workspace that grows with the
 * batch).  The call makes no hipMalloc / hipFree and nothing else that allocates device memory; the first
 * call for a shape and filter length also queries that kernel's occupancy, so make one call before a
 * capture.  All calls on one GPU share the array (whatever context or stream they come through).  Calls
 * through one context are ordered by its stream.  A call through another context of the same GPU first
 * makes its stream wait (hipStreamWaitEvent on an event made in wn_create; no allocation) for the last
 * call that used the array, and a mutex orders host threads, so calls issued through this entry point
 * never overlap on the array.  What remains the caller's: a context whose stream is being captured must
 * be the only one calling on that GPU during the capture, and replays of a captured graph are ordered
 * against other calls by the caller.  Nothing in the array has to be zero beforehand (every slot that is
 * read was written by the same call).
 */
#define WN_STATIC_SLOTS 8192
int wn_train_batch_static(wn_context* ctx, const double* X_dev, int B, int R, int C, double* sums_dev);

/*
 * Device pointer to the N + 2 sums of the last wn_train_batch on this context, for the caller's
 * all-reduce over data-parallel ranks (sum), e.g. torch.distributed / RCCL on the same stream.
 */
int wn_sums(wn_context* ctx, double** sums_dev, int* count);
int wn_get_sums(wn_context* ctx, double* sums_host, int* count); /* synchronises */
/*
 * Makes the context keep its sums in caller-owned device memory (at least WN_MAX_TAPS + 2
 * doubles), e.g. a framework tensor the caller's collective library can reduce directly.
 * NULL switches back to the context's own buffer.
 */
int wn_set_sums_buffer(wn_context* ctx, double* sums_dev);

/*
 * Wavenet::flushBatchQueue_ + update_ + setFilter (src/Wavenet.cxx:717-765, 8-36) on the
 * device, from the (all-reduced) sum buffer:
 *   gbar = (sum_g + lambda_sum * RegTermDeriv(a)) / float(batch_total)
 *   p    = p * inertia_eff;  p += -alpha * gbar;  a = a + p
 *   cost = (sum_S + lambda_sum * RegTerm(a_old)) / float(batch_total)
 * lambda_sum is the sum over the batch's examples of lambda (= batch_total * lambda unless
 * Coach's annealing changes it per event, src/Coach.cxx:176-180); inertia_eff is the caller's
 * evaluation of src/Wavenet.cxx:757.  The new filter and the cost are appended to device-side
 * logs (Wavenet::filterLog / costLog).
 * A batch whose sums are not finite is NOT applied, as the reference's train() fails before it
 * queues anything (src/Wavenet.cxx:234-241): filter, momentum and logs stay as they were, every
 * later wn_apply_update is skipped too, and wn_num_updates reports WN_ERR_NONFINITE until
 * wn_set_filter or wn_clear_logs.  The call itself is asynchronous and returns WN_OK.
 */
int wn_apply_update(wn_context* ctx, double batch_total, double lambda_sum, double alpha,
                    double inertia_eff, int do_wavelet);

/*
 * One whole batch of Coach's loop in one submission: wn_train_batch (without per-example outputs), the
 * all-reduce of the sums when the context has a communicator of more than one rank (wn_comm_init), and
 * wn_apply_update -- src/Wavenet.cxx:184-242 for B examples followed by flushBatchQueue_ (:717-765).
 * Same arithmetic, same results as the three calls.  For 64x64 examples the batch reduction (and,
 * without a communicator, the update) runs inside the training kernel, in the workgroup that finishes
 * last, so a step is ONE kernel launch; for the other shapes of up to WN_MAX_FUSED samples it is the
 * training kernel plus one launch that reduces and updates.  Either way consecutive steps queue up on
 * the stream with no host code between them.  batch_total is the batch size over all ranks, B this
 * rank's share of it.
 */
int wn_train_step(wn_context* ctx, const double* X_dev, int B, int R, int C, double batch_total,
                  double lambda_sum, double alpha, double inertia_eff, int do_wavelet);

/* Number of updates applied since the context was created or wn_clear_logs (wn_set_filter does not
 * reset it).  Returns WN_ERR_NONFINITE (with *steps filled in) once a batch was rejected for
 * non-finite sums.  Synchronises. */
int wn_num_updates(wn_context* ctx, int* steps);
/*
 * Copies log entries [first, first+count): filters_host[count][N] are the filters after each
 * update, costs_host[count] the batch-mean costs.  Either may be NULL.  Synchronises.
 */
int wn_get_logs(wn_context* ctx, int first, int count, double* filters_host, double* costs_host);
/*
 * Everything the host side of Wavenet::train / flushBatchQueue_ (src/Wavenet.cxx:219-242, 717-765) keeps after a
 * run of batches -- m_filterLog and m_costLog entries, m_momentum, and whether an update was refused -- in ONE
 * readback: *steps = number of updates applied so far; the log entries [first, min(first + max_count, *steps)) as
 * wn_get_logs gives them; momentum_host[N].  Any of the three arrays may be NULL.  Returns WN_ERR_NONFINITE, with
 * all outputs filled in, once a batch was rejected for non-finite sums.  Synchronises.
 */
int wn_get_updates(wn_context* ctx, int first, int max_count, int* steps, double* filters_host, double* costs_host,
                   double* momentum_host);
int wn_clear_logs(wn_context* ctx);

/*
 * Wavenet::costMap (src/Wavenet.cxx:278-324) evaluates cost(forward(X)) on a grid of filters, one
 * setFilter + forward + SparseTerm per grid point and example.  This is the batched form of its
 * device part: S_dev[f*B + e] = SparseTerm(forward(example e)) under filter f, for F filters of N
 * taps each given row by row in filters_host.  The context's own filter, momentum and logs are not
 * touched (none needs to be set); the regularisation term of each filter is wn_reg_term.  Examples
 * of up to WN_MAX_FUSED samples; synchronises before returning.
 */
int wn_sparse_cost_grid(wn_context* ctx, const double* X_dev, int B, int R, int C,
                        const double* filters_host, int F, int N, double* S_dev);

/* ---- generators --------------------------------------------------------------------------- */
/*
 * Batches of synthetic examples written straight into device memory, in the layout of every other
 * entry point, standing for B calls of GeneratorBase::next (include/Wavenet/GeneratorBase.h:32-103).
 * The reference's generators reseed from entropy on every reset (include/Wavenet/Generators.h:66, 100,
 * 170) and cannot be reproduced; these are counter-based -- every value is a function of (seed,
 * example number, item) -- and bit-identical to wavenet_amd/generators.py and the host layer's
 * generators.  Example e of the call has number first_example + e, so consecutive calls (or the
 * ranks of a data-parallel job) continue one sequence.  At most 65535 examples per call.
 *   uniform      U[0,1) pixels (UniformGenerator, Generators.h:87-96)
 *   needle       the same with everything below the threshold zeroed, one pixel kept in an image
 *                that would be empty (NeedleGenerator, Generators.h:50-62)
 *   calorimeter  eta-phi image in the spirit of HepMCGenerator's TH2F fill (Generators.h:377-393):
 *                K deposits with K from poisson_table_host[4096] (inverse CDF samples) and energies
 *                from energy_table_host[1024], plus `jets` clusters of jet_deposits deposits; bins
 *                rounded to float.  The tables come from the caller so that host and device agree
 *                to the bit; the call synchronises to copy them.
 */
int wn_generate_uniform(wn_context* ctx, unsigned long long seed, long long first_example,
                        int B, int R, int C, double* X_dev);
int wn_generate_needle(wn_context* ctx, unsigned long long seed, long long first_example,
                       int B, int R, int C, double threshold, double* X_dev);
int wn_generate_calorimeter(wn_context* ctx, unsigned long long seed, long long first_example,
                            int B, int R, int C, const int* poisson_table_host,
                            const double* energy_table_host, int jets, int jet_deposits, double* X_dev);
/*
 * GaussianGenerator (include/Wavenet/Generators.h:122-166): one to three unit-height Gaussian bumps of random
 * centre and width per example.  exp_table_host[i] = exp(-i / 128), i = 0 .. 4096 (4097 entries): the bumps are
 * evaluated from it by linear interpolation, with the same arithmetic as the host layer's GaussianGenerator, so
 * that the two agree to the bit whatever the libm.  The table is kept on the device until a different one is passed.
 */
int wn_generate_gaussian(wn_context* ctx, unsigned long long seed, long long first_example, int B, int R, int C,
                         const double* exp_table_host, double* X_dev);

/* ---- measurement --------------------------------------------------------------------------- */
/*
 * Brackets the training kernel of every following wn_train_batch with a pair of HIP events on
 * the context's stream (at most max_records launches; 0 turns it off).  For examples that take the
 * streamed path the pair spans all of its kernels.  wn_profile_collect synchronises and returns the
 * durations in milliseconds, oldest first.
 */
int wn_profile_begin(wn_context* ctx, int max_records);
int wn_profile_collect(wn_context* ctx, float* ms_host, int capacity, int* count);

/* ---- data-parallel exchange (RCCL over xGMI) ---------------------------------------------- */
/*
 * One communicator per context / process.  Rank 0 calls wn_comm_unique_id and ships the 128
 * bytes to the other ranks by any means; then every rank calls wn_comm_init.
 * wn_allreduce_sums sums the context's N + 2 doubles over all ranks in place, on the stream.
 */
int wn_comm_unique_id(void* id128_host);
int wn_comm_init(wn_context* ctx, const void* id128_host, int nranks, int rank);
int wn_comm_destroy(wn_context* ctx);
int wn_allreduce_sums(wn_context* ctx);
/*
 * The same for ONE process that drives several GPUs from one host thread (the host layer's data-parallel
 * Coach): communicators for all n contexts at once (every context on a GPU of its own), and the all-reduce
 * of all of them as one group, each on its context's stream.
 */
int wn_comm_init_all(wn_context** ctxs, int n);
int wn_allreduce_sums_all(wn_context** ctxs, int n);

#ifdef __cplusplus
}
#endif
#endif /* WAVENET_HIP_H */
