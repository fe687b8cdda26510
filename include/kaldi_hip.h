/* kaldi_hip.h -- C ABI of libkaldi_hip.so, the MI355X (gfx950) device layer of the nnet1 hot path.
 *
 * It replaces, for one GPU per process, everything the reference's cudamatrix layer obtained
 * from outside its own C++:
 *   - the kernel wrappers of   src/cudamatrix/cu-kernels-ansi.h:33-94   (cudaI32_* / cudaF_*)
 *   - the RNG wrappers of      src/cudamatrix/cu-randkernels-ansi.h:33-35
 *   - cuBLAS sgemm             src/cudamatrix/cu-matrix-inl.h:860-907
 *   - the CUDA runtime calls   src/cudamatrix/cu-matrix-inl.h:96-310, cu-vector-inl.h:64-200,
 *                              cu-stlvector-inl.h:64-145, cu-device.cc:56-210
 *
 * Conventions (differences from the reference ABI are deliberate):
 *   - every entry point returns int: 0 on success, otherwise a hipError_t value (or one of the
 *     KH_ERR_* codes below); hipK_error_string() turns it into text.  The reference returned
 *     void and polled cudaGetLastError() afterwards (cu-common.h:35-42).
 *   - no `dim3 Gr, dim3 Bl`: launch geometry is chosen inside the library for CDNA4.
 *   - the first argument is the stream (a hipStream_t passed as void*; NULL = the library's
 *     default stream, see hipK_default_stream()).  Nothing synchronises implicitly except the
 *     functions documented as blocking.
 *   - all data pointers are device pointers owned by the caller unless the name says `host`.
 *   - matrices are row-major; element (r,c) of a matrix with MatrixDim d lives at r*d.stride+c
 *     (cu-matrixdim.h:41-45).  Strides are in elements.
 *   - only float (`hipF_`) and int32 (`hipI32_`) are provided: no file of the reference's nnet1
 *     path instantiates CuMatrix<double> (BaseFloat = float, base/kaldi-types.h:27).
 *   - arithmetic notes that matter for parity are written at each entry point.
 *   - threading: like the reference's cudamatrix layer (one host thread, a singleton device), the
 *     library is meant to be driven by one host thread; its state (default stream, per-stream
 *     workspaces, the pinned staging buffers of the blocking copies, the last error message) is not
 *     locked.  The exceptions, callable from any thread beside that one: hipK_event_synchronize and
 *     hipK_stream_synchronize on an event / a stream the caller owns -- they touch no library state on
 *     success and are what a writer thread needs to wait for a copy queued by the driving thread
 *     (nnet-forward's posterior writer does exactly and only that).
 */
#ifndef KALDI_AMD_INCLUDE_KALDI_HIP_H_
#define KALDI_AMD_INCLUDE_KALDI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same layout as the reference's MatrixDim (cudamatrix/cu-matrixdim.h:41-45) */
#ifndef KALDI_AMD_MATRIXDIM_DEFINED
#define KALDI_AMD_MATRIXDIM_DEFINED
typedef struct MatrixDim_ {
  int32_t rows;
  int32_t cols;
  int32_t stride;
} MatrixDim;
#endif

typedef void *khStream_t; /* hipStream_t */

enum {
  KH_OK = 0,
  KH_ERR_INVALID_ARG = 100001, /* shape / pointer check failed before any launch */
  KH_ERR_NOT_INITIALISED = 100002,
  KH_ERR_COMM = 100003,        /* RCCL failure, see hipK_error_string */
  KH_ERR_TOO_LARGE = 100004
};

enum { KH_NO_TRANS = 111, KH_TRANS = 112 }; /* values of kNoTrans / kTrans (matrix/matrix-common.h) */

/* ------------------------------------------------------------------------------------------
 * Device / runtime   (replaces cu-device.cc:56-210 and the cudaMalloc/cudaMemcpy call sites)
 * ------------------------------------------------------------------------------------------ */

/* Select a GPU and create the library's default stream.  device_id < 0 picks the visible GPU
 * with the most free memory (cu-device.cc:62-118); otherwise that ordinal (SelectGpuId,
 * cu-device.cc:136-175).  Idempotent for the same device. */
int hipK_init(int device_id);
int hipK_shutdown(void);
int hipK_active_device(int *device_id);
int hipK_device_count(int *count);
int hipK_mem_info(size_t *free_bytes, size_t *total_bytes); /* cuMemGetInfo, cu-device.cc:210 */
int hipK_default_stream(khStream_t *stream);
int hipK_stream_create(khStream_t *stream);
int hipK_stream_destroy(khStream_t stream);
int hipK_stream_synchronize(khStream_t stream); /* blocking */
int hipK_device_synchronize(void);              /* blocking */
const char *hipK_error_string(int status);
/* compute units, wave size, gcn arch name ("gfx950...") of the active device */
int hipK_device_props(int *num_cu, int *wave_size, char *arch_name, size_t arch_name_len);

/* Events for timing on a stream (the reference timed with host Timer + device sync,
 * cu-matrix-inl.h:349-357; events do not stall the stream). */
typedef void *khEvent_t; /* hipEvent_t */
int hipK_event_create(khEvent_t *ev);
/* an event that only orders streams (hipK_stream_wait_event): no time stamps are taken when it is recorded */
int hipK_event_create_ordering(khEvent_t *ev);
int hipK_event_destroy(khEvent_t ev);
int hipK_event_record(khEvent_t ev, khStream_t s);
int hipK_event_synchronize(khEvent_t ev); /* blocking */
/* Work queued on `s` after this call starts only when everything recorded into `ev` has finished
 * (no host blocking).  No counterpart in the reference, which used one stream: it orders the
 * weight updates queued on a second stream against the backward pass. */
int hipK_stream_wait_event(khStream_t s, khEvent_t ev);
int hipK_event_elapsed_ms(float *ms, khEvent_t start, khEvent_t stop);

/* cudaMallocPitch (cu-matrix-inl.h:108): rows x cols floats, row pitch rounded up so that every
 * row starts 256-byte aligned; *stride_elems receives the pitch in elements.  Not zeroed. */
int hipK_malloc_pitch(void **ptr, int32_t *stride_elems, int32_t rows, int32_t cols, size_t elem_size);
int hipK_malloc(void **ptr, size_t bytes);          /* cudaMalloc, cu-vector-inl.h:74 */
int hipK_free(void *ptr);
int hipK_malloc_host(void **ptr, size_t bytes);     /* pinned host staging buffer */
int hipK_free_host(void *ptr);
int hipK_memset_async(khStream_t s, void *ptr, int value, size_t bytes); /* cudaMemset */
/* cudaMemcpy2D (cu-matrix-inl.h:155-284).  kind: 0 h2d, 1 d2h, 2 d2d.  Pitches and width in bytes.
 * h2d/d2h block until the copy has completed; d2d is asynchronous on the stream. */
int hipK_memcpy_2d(khStream_t s, void *dst, size_t dpitch, const void *src, size_t spitch,
                   size_t width_bytes, size_t height, int kind);
int hipK_memcpy(khStream_t s, void *dst, const void *src, size_t bytes, int kind);
/* The same 2-D copy, only queued on `s`: the host does not wait.  The host side must be pinned memory from
 * hipK_malloc_host and belongs to the copy until an event recorded on `s` after this call has completed.  The
 * reference's copies all block (cu-matrix-inl.h:173-222); nnet-forward uses this one to bring the posteriors
 * of one utterance to the host while the next utterance is computed (nnetbin/nnet-forward.cc:127-160). */
int hipK_memcpy_2d_async(khStream_t s, void *dst, size_t dpitch, const void *src, size_t spitch,
                         size_t width_bytes, size_t height, int kind);

/* ------------------------------------------------------------------------------------------
 * int32 kernels
 * ------------------------------------------------------------------------------------------ */
/* cudaI32_set_const (cu-kernels-ansi.h:33) */
int hipI32_set_const(khStream_t s, int32_t *mat, int32_t value, MatrixDim d);

/* ------------------------------------------------------------------------------------------
 * float kernels: CuMatrix   (cu-kernels-ansi.h:44-67; kernels cu-kernels.cu:120-454)
 * ------------------------------------------------------------------------------------------ */
int hipF_set_const(khStream_t s, float *mat, float value, MatrixDim d);
/* counts[0] = number of NaN elements, counts[1] = number of +infinity elements of mat (counts: 2 int32 in
 * device memory, zeroed and filled on `s`).  The reference scans every output matrix for exactly these two on
 * the host, after the copy (nnetbin/nnet-forward.cc:147-156); here the scan runs where the matrix already is. */
int hipF_count_nonfinite(khStream_t s, const float *mat, MatrixDim d, int32_t *counts);
int hipF_part_set_const(khStream_t s, float *mat, float value, MatrixDim d, int32_t ro, int32_t r,
                        int32_t co, int32_t c);
int hipF_binarize(khStream_t s, float *mat, float thres, MatrixDim d);
int hipF_add_const(khStream_t s, float *mat, float value, MatrixDim d);
int hipF_power(khStream_t s, float *mat, float pow, MatrixDim d);
int hipF_round(khStream_t s, float *mat, MatrixDim d);
int hipF_scale(khStream_t s, float *mat, float value, MatrixDim d);
int hipF_apply_floor(khStream_t s, float *mat, float value, MatrixDim d);
int hipF_apply_truncate(khStream_t s, float *mat, float low, float high, MatrixDim d);
int hipF_apply_log(khStream_t s, float *mat, MatrixDim d);
int hipF_apply_exp(khStream_t s, float *mat, MatrixDim d);
/* mat and A share d (the reference asserts equal stride, cu-matrix-inl.h:545) */
int hipF_mul_elements(khStream_t s, float *mat, const float *A, MatrixDim d);
int hipF_div_elements(khStream_t s, float *mat, const float *A, MatrixDim d);
int hipF_mul_cols_vec(khStream_t s, float *mat, const float *scale, MatrixDim d);
int hipF_mul_rows_vec(khStream_t s, float *mat, const float *scale, MatrixDim d);
/* multiply by the float reciprocal 1.0f/v, as the reference kernels do (cu-kernels.cu:304,326) */
int hipF_div_cols_vec(khStream_t s, float *mat, const float *vec_div, MatrixDim d);
int hipF_div_rows_vec(khStream_t s, float *mat, const float *vec_div, MatrixDim d);
/* dst = alpha*A + beta*dst; A and dst share d.  Also serves CuVector::AddVec (rows = 1). */
int hipF_add_mat(khStream_t s, float alpha, const float *A, float beta, float *dst, MatrixDim d);
int hipF_part_add_mat(khStream_t s, float alpha, const float *A, MatrixDim da, int32_t ro, int32_t co,
                      float beta, float *dst, MatrixDim d);
int hipF_add_vec_to_cols(khStream_t s, float alpha, const float *col, float beta, float *dst, MatrixDim d);
int hipF_add_vec_to_rows(khStream_t s, float alpha, const float *row, float beta, float *dst, MatrixDim d);
int hipF_add_vec_to_partial_rows(khStream_t s, float alpha, int32_t offset, const float *row, int32_t dim,
                                 float beta, float *dst, MatrixDim d);
/* dst = log(exp(A)+exp(dst)), evaluated in double like cu-kernels.cu:362-380 */
int hipF_log_add_exp_mat(khStream_t s, const float *A, float *dst, MatrixDim d);
/* rows whose L2 norm exceeds bound are rescaled to it (cu-kernels.cu:431-454; double accumulation) */
int hipF_apply_row_l2upperbound(khStream_t s, float *mat, float bound, MatrixDim d);

/* ------------------------------------------------------------------------------------------
 * float kernels: CuVector   (cu-kernels-ansi.h:71-73; host slab loops cu-vector-inl.h:443-529)
 * The reference launched one 256-row (256-column) slab per call and looped on the host; here one
 * call reduces the whole matrix:  vec = alpha * sum + beta * vec.
 * ------------------------------------------------------------------------------------------ */
/* vec[c] = alpha * sum_r mat[r,c] + beta * vec[c]   (CuVector::AddRowSumMat).  fp32 accumulation. */
int hipF_add_row_sum_mat(khStream_t s, float alpha, const float *mat, MatrixDim d, float beta, float *vec);
/* vec[r] = alpha * sum_c mat[r,c] + beta * vec[r]   (CuVector::AddColSumMat) */
int hipF_add_col_sum_mat(khStream_t s, float alpha, const float *mat, MatrixDim d, float beta, float *vec);
int hipF_invert_elements(khStream_t s, float *data, MatrixDim d);

/* ------------------------------------------------------------------------------------------
 * float kernels: cu:: functions   (cu-kernels-ansi.h:78-94)
 * ------------------------------------------------------------------------------------------ */
int hipF_relu(khStream_t s, float *y, const float *x, MatrixDim d);
int hipF_diff_relu(khStream_t s, float *eout, const float *e, const float *y, MatrixDim d);
int hipF_softrelu(khStream_t s, float *y, const float *x, MatrixDim d);
int hipF_diff_softrelu(khStream_t s, float *eout, const float *e, const float *x, MatrixDim d);
/* y = 1/(1+exp(-x)) (cu-kernels.cu:577-585).  y, x (and eout, e, y below) share d. */
int hipF_sigmoid(khStream_t s, float *y, const float *x, MatrixDim d);
/* eout = y*(1-y)*e (cu-kernels.cu:590-596) */
int hipF_diff_sigmoid(khStream_t s, float *eout, const float *e, const float *y, MatrixDim d);
/* Row softmax with max subtraction in ONE pass over HBM: y[r,:] = exp(x[r,:]-max_r)/sum.
 * Replaces the whole of cu::Softmax (cu-math-inl.h:209-251: FindRowMaxId + softmax_part +
 * AddColSumMat + DivRowsVec) as well as the disabled cuda_softmax (cu-kernels.cu:601-621).
 * The sum is accumulated in fp32 (like CPU ApplySoftMax, matrix/kaldi-vector.cc:678-685) and
 * every element is multiplied by the fp32 reciprocal of the sum.  y may alias x. */
int hipF_softmax(khStream_t s, float *y, const float *x, MatrixDim d);
/* Y = exp(X - X[r, vec_ids[r]])  (cu-kernels.cu:751-759); kept for callers that build softmax from parts */
int hipF_softmax_part(khStream_t s, const float *X, const int32_t *vec_ids, float *Y, MatrixDim d);
int hipF_regularize_l1(khStream_t s, float *wei, float *grad, float l1, float lr, MatrixDim d);
/* vec_id[r] = argmax_c mat[r,c]; vec_val[r] (may be NULL) = that maximum.  One call covers all
 * columns (the reference looped over 256-column slabs, cu-math-inl.h:297-332).  Ties resolve to the
 * LOWEST column index, which is what the reference's CPU path does (strict '<', cu-math-inl.h:343);
 * a row whose elements are all <= -1e21 (or NaN) yields -1, as there. */
int hipF_find_row_max_id(khStream_t s, const float *mat, float *vec_val, int32_t *vec_id, MatrixDim d);
/* vec_log_post[r] = log(max(mat[r,tgt[r]], 1e-20)); mat[r,tgt[r]] -= 1  (cu-kernels.cu:733-745) */
int hipF_diff_xent(khStream_t s, const int32_t *vec_tgt, float *mat_net_out, float *vec_log_post, MatrixDim d);
/* y[r,:] = x[copy_from[r],:] for r < d_out.rows (cu-kernels.cu:661-669).  d_in describes all of x:
 * an index outside [0, d_in.rows) leaves that output row untouched instead of reading out of bounds. */
int hipF_randomize(khStream_t s, float *y, const float *x, const int32_t *copy_from, MatrixDim d_out,
                   MatrixDim d_in);
/* splice: y[r, k*D+c] = x[clamp(r+off[k], 0, rows-1), c]  (cu-kernels.cu:627-638) */
int hipF_expand(khStream_t s, float *y, const float *x, const int32_t *off, MatrixDim d_out, MatrixDim d_in);
/* column gather: y[r,c] = x[r, copy_from[c]]; an index outside [0, d_in.cols) writes +inf
 * (cu-kernels.cu:644-656) */
int hipF_copy(khStream_t s, float *y, const float *x, const int32_t *copy_from, MatrixDim d_out, MatrixDim d_in);

/* ------------------------------------------------------------------------------------------
 * RNG   (cu-randkernels-ansi.h:33-35; kernels cu-randkernels.cu:44-120)
 * Per-element HybridTaus state in four uint32 arrays laid out like the matrix (rows*stride);
 * bit-for-bit the reference generator, so equal seeds give equal draws.
 * ------------------------------------------------------------------------------------------ */
int hipF_rand(khStream_t s, float *mat, uint32_t *z1, uint32_t *z2, uint32_t *z3, uint32_t *z4, MatrixDim d);
int hipF_gauss_rand(khStream_t s, float *mat, uint32_t *z1, uint32_t *z2, uint32_t *z3, uint32_t *z4, MatrixDim d);
/* states = probs > rand ? 1 : 0 */
int hipF_binarize_probs(khStream_t s, float *states, const float *probs, const float *rand, MatrixDim d);
/* Fused forms used by CuRand: draw and apply in one pass, state updated in place, no temporary
 * matrix.  Same draws as hipF_rand / hipF_gauss_rand followed by binarize_probs / add_mat
 * (cu-rand-inl.h:139-177). */
int hipF_binarize_probs_rand(khStream_t s, float *states, const float *probs, uint32_t *z1, uint32_t *z2,
                             uint32_t *z3, uint32_t *z4, MatrixDim d);
int hipF_add_gauss_noise(khStream_t s, float *mat, float gscale, uint32_t *z1, uint32_t *z2, uint32_t *z3,
                         uint32_t *z4, MatrixDim d);

/* ------------------------------------------------------------------------------------------
 * GEMM   (replaces cublasSgemm at cu-matrix-inl.h:860-907)
 * C(MxN) = alpha * op(A) * op(B) + beta * C, row-major, leading dimensions in elements.
 * op(A) is MxK, op(B) is KxN.  fp32 in, fp32 accumulate on the matrix cores
 * (v_mfma_f32_32x32x2_f32); products are summed in a fixed, run-to-run reproducible order.
 * beta == 0 never reads C (so C may hold NaNs).
 * ------------------------------------------------------------------------------------------ */
int hipF_gemm(khStream_t s, int transA, int transB, int32_t M, int32_t N, int32_t K, float alpha,
              const float *A, int32_t lda, const float *B, int32_t ldb, float beta, float *C, int32_t ldc);

/* Measurement aid: with timing enabled every GEMM main-kernel launch (hipF_gemm and the fused
 * affine / rbm entry points) carries two HIP events BOUND to the dispatch (they read the kernel's own begin
 * and end stamps, as a kernel trace does, and put no packet of their own into the stream).  hipK_gemm_timing_read
 * blocks until they have completed, returns the summed kernel time, the time during which at
 * least one of them was running (launches queued on two streams overlap, so this is less than the
 * sum), the number of launches and their algorithmic FLOPs (2*M*N*K each), and clears the record.
 * Off by default. */
int hipK_gemm_timing(int enable);
int hipK_gemm_timing_read(double *total_ms, double *busy_ms, long long *launches, double *flops);
/* The launches recorded since the last hipK_gemm_timing_read, in launch order, without clearing them: start
 * (ms after the first launch's start), duration (ms) and algorithmic GFLOP of each; at most `cap` entries,
 * *count receives how many were written.  Blocks until they have completed. */
int hipK_gemm_timing_spans(double *start_ms, double *dur_ms, double *gflop, int cap, int *count);

/* Tuning aid: pin the tile (bm x bn) and the split-K factor of every following GEMM instead of
 * letting the library choose; zeros restore the automatic choice.  Also settable at start-up with
 * the environment variable KALDI_HIP_GEMM_FORCE="bm,bn,splitk". */
int hipK_gemm_force_plan(int bm, int bn, int splitk);

/* on != 0: every following GEMM sums each output element's K range in one piece, in K order (no split of K over
 * workgroups or over the waves of a workgroup), whatever the number of rows: a row of the result then has the same
 * bits whether it is computed alone or together with other rows.  For callers that batch independent rows
 * (utterances) into one call and promise results that do not depend on the batching.  Off by default: the split
 * plans are the faster ones for short products. */
int hipK_gemm_row_invariant(int on);

/* Pre-sizes the scratch workspace of the library's default stream (split-K slabs of hipF_gemm and of the
 * fused entry points).  The library grows the workspace on demand (blocking: device synchronise + free +
 * allocate), so callers only need this to keep that out of a timed or captured region. */
int hipK_reserve_workspace(size_t bytes);

/* ------------------------------------------------------------------------------------------
 * Fused entry points of the nnet1 hot path.  Each one is arithmetic-for-arithmetic the sequence
 * of reference calls named in its comment; they exist so that the activations make one trip
 * through HBM instead of three or four.
 * ------------------------------------------------------------------------------------------ */
enum { KH_ACT_NONE = 0, KH_ACT_SIGMOID = 1, KH_ACT_SOFTMAX = 2 };

/* out(NxO) = act( bias(row, O) + in(NxI) * W(OxI)^T )
 *   = BiasedLinearity::PropagateFnc (nnet-biasedlinearity.h:58-63) [+ Sigmoid::PropagateFnc
 *     (nnet-activation.h:207-210) | Softmax::PropagateFnc (:233-236)]
 *   = Rbm::PropagateFnc (nnet-rbm.h:136-145)
 * trans_w = KH_NO_TRANS computes in * W instead (W is IxO): Rbm::Reconstruct (nnet-rbm.h:158-176). */
int hipF_affine_forward(khStream_t s, int act, int trans_w, int32_t N, int32_t O, int32_t I, const float *in,
                        int32_t ldin, const float *W, int32_t ldw, const float *bias, float *out, int32_t ldout);

/* eout(NxI) = e(NxO) * W(OxI) [ .* y .* (1-y) when y != NULL ]
 *   = BiasedLinearity::BackpropagateFnc (nnet-biasedlinearity.h:65-69) followed by
 *     Sigmoid::BackpropagateFnc of the layer below (nnet-activation.h:212-218). */
int hipF_affine_backward(khStream_t s, int32_t N, int32_t O, int32_t I, const float *e, int32_t lde,
                         const float *W, int32_t ldw, const float *y, int32_t ldy, float *eout, int32_t ldeout);

/* BiasedLinearity::Update (nnet-biasedlinearity.h:71-102) without the L1 / max-norm options:
 *   corr  = gscale * e^T * in + momentum * corr        (corr OxI; gscale = 1 or 1/N)
 *   bcorr = gscale * colsum(e) + momentum * bcorr
 *   W     = (1 - l2) * W - lr * corr                   (l2 = lr * l2_penalty * N, 0 to disable)
 *   b     = b - lr * bcorr
 * The weight pass is the epilogue of the gradient GEMM: W and corr are read and written once.
 * corr / bcorr may be NULL only when momentum == 0 (the gradient is then not stored). */
int hipF_affine_update(khStream_t s, int32_t N, int32_t O, int32_t I, const float *e, int32_t lde,
                       const float *in, int32_t ldin, float gscale, float momentum, float lr, float l2,
                       float *W, int32_t ldw, float *corr, int32_t ldcorr, float *bias, float *bcorr);

/* hipF_affine_update with the new weights written to W_out (W_in itself, or a second O x I buffer as in
 * hipF_affine_update_weights): weight pass and bias pass in ONE launch -- the bias sums are taken by a few
 * extra workgroups at the start of the gradient GEMM's grid.  Same bits as the two halves below. */
int hipF_affine_update_fused(khStream_t s, int32_t N, int32_t O, int32_t I, const float *e, int32_t lde,
                             const float *in, int32_t ldin, float gscale, float momentum, float lr, float l2,
                             const float *W_in, int32_t ldw_in, float *W_out, int32_t ldw_out, float *corr,
                             int32_t ldcorr, float *bias, float *bcorr);

/* hipF_affine_forward(KH_ACT_SIGMOID, KH_TRANS, ...) and hipF_binarize_probs_rand on its output in ONE launch -- the
 * positive phase of an RBM CD-1 step with Bernoulli hidden units (Rbm::Propagate, nnet-rbm.h:136-145, followed by
 * CuRand::BinarizeProbs, cu-rand-inl.h:139-169): probs = sigmoid(in * W^T + b), states = (probs > u) ? 1 : 0 with one
 * HybridTaus draw u per element from the state arrays z1..z4, which are indexed like probs (r * ldprobs + c) and
 * advanced exactly as hipF_binarize_probs_rand advances them: same probabilities, same states, same generator
 * state afterwards as the two calls.  ldstates must equal ldprobs. */
int hipF_affine_forward_sample(khStream_t s, int32_t N, int32_t O, int32_t I, const float *in, int32_t ldin,
                               const float *W, int32_t ldw, const float *bias, float *probs, int32_t ldprobs,
                               float *states, int32_t ldstates, uint32_t *z1, uint32_t *z2, uint32_t *z3, uint32_t *z4);

/* Several dependent layers of a forward pass in ONE launch: layer i computes out_i = act(in_i * W_i^T + b_i) exactly
 * as hipF_affine_forward(act, KH_TRANS, ...) would -- same kernels' tile code, same bits -- with in_i = out_{i-1}
 * for i > 0.  The workgroups hand the activations of a row block to each other inside the launch (write-through
 * stores, a counter per layer and row block, loads past the L1) instead of through a kernel boundary.  Taken only
 * where every layer is the one-workgroup-per-compute-unit shape of a 256-frame bunch (same tile grid in every layer,
 * at most one workgroup per compute unit, K a whole number of tiles, 16-byte aligned operands); anything else is
 * run as `count` hipF_affine_forward launches by the call itself.  act: KH_ACT_NONE or KH_ACT_SIGMOID.  The
 * library holds one set of hand-off counters per device: chains must not overlap in time on one device. */
typedef struct {
  int32_t O, I;
  const float *in; int32_t ldin;
  const float *W; int32_t ldw;
  const float *bias;
  float *out; int32_t ldout;
} khAffineForward;
int hipF_affine_forward_chain(khStream_t s, int act, int32_t N, int32_t count, const khAffineForward *layers);

/* The updates of several layers in ONE launch: item i is exactly one hipF_affine_update_fused call (same
 * arguments, same bits), the items must not overlap in what they write.  For a training step that queues the
 * updates of all its <biasedlinearity> layers behind the back-propagation pass (each BiasedLinearity::Update,
 * nnet-biasedlinearity.h:71-102, needs only its layer's input, its output error and its own weights): a
 * 256-frame update is short, so launched one by one every layer pays a kernel boundary and all its
 * workgroups load, multiply and store in step; in one long grid the workgroups of a compute unit drift out
 * of phase and one's loads and stores run under another's MFMAs.  Items that do not qualify for the
 * grouped kernel (operands not 16-byte aligned, more than 2 GiB) are launched on their own.  Every item is
 * checked before anything is launched: a call that returns an argument error has changed nothing. */
typedef struct {
  int32_t N, O, I;
  const float *e; int32_t lde;
  const float *in; int32_t ldin;
  float gscale, momentum, lr, l2;
  const float *W_in; int32_t ldw_in;
  float *W_out; int32_t ldw_out;
  float *corr; int32_t ldcorr;
  float *bias; float *bcorr;
} khAffineUpdate;
int hipF_affine_update_group(khStream_t s, int32_t count, const khAffineUpdate *items);

/* The first half of the data-parallel update (no counterpart in the reference, which trains on one GPU) for
 * several layers in ONE launch of the same grouped kernel: item i writes the raw sums of its layer,
 *   gW = e^T * in   (O x I, leading dimension ldg; the floats between I and ldg of a row are left alone)
 *   gb = colsum(e)  (O floats),
 * which the caller exchanges between the ranks and then applies (hipF_sgd_apply_segments, hipF_affine_apply).
 * K = the bunch is summed in one piece in a fixed order: the same bits whichever items share the launch.
 * Items that do not qualify for the grouped kernel (operands or gW not 16-byte aligned, I or ldg not a multiple
 * of 4, more than 2 GiB) are computed on their own by hipF_gemm and hipF_add_row_sum_mat. */
typedef struct {
  int32_t N, O, I;
  const float *e; int32_t lde;
  const float *in; int32_t ldin;
  float *gW; int32_t ldg;
  float *gb;
} khAffineGradient;
int hipF_affine_gradient_group(khStream_t s, int32_t count, const khAffineGradient *items);
/* The same, and `done` (an event of hipK_event_create_ordering; may be NULL) is recorded behind the call's work: bound
 * to the call's last launch as that launch's own completion signal where possible, so that the stream's queue holds
 * no separate mark for the next launch to wait behind. */
int hipF_affine_gradient_group_marked(khStream_t s, int32_t count, const khAffineGradient *items, khEvent_t done);

/* The two halves of hipF_affine_update on their own, so that a caller can queue them on different
 * streams (the bias half reads only e; the weight half reads only e and in besides W):
 *   weights: corr = gscale * e^T * in + momentum * corr ; W_out = (1 - l2) * W_in - lr * corr
 *   bias   : bcorr = gscale * colsum(e) + momentum * bcorr ; b = b - lr * bcorr
 * W_out may be W_in (in place) or a second O x I buffer: reading the old weights from one buffer
 * and writing the new ones to another lets the step run while other work still reads the old
 * weights (the reference updates in place, nnet-biasedlinearity.h:95-101; the values are the same). */
int hipF_affine_update_weights(khStream_t s, int32_t N, int32_t O, int32_t I, const float *e, int32_t lde,
                               const float *in, int32_t ldin, float gscale, float momentum, float lr, float l2,
                               const float *W_in, int32_t ldw_in, float *W_out, int32_t ldw_out, float *corr,
                               int32_t ldcorr);
int hipF_affine_update_bias(khStream_t s, int32_t N, int32_t O, const float *e, int32_t lde, float gscale,
                            float momentum, float lr, float *bias, float *bcorr);

/* Data-parallel form of the update, second half (the first half is a plain hipF_gemm into a dense
 * gradient buffer plus hipF_add_row_sum_mat, all-reduced across ranks by the caller):
 *   corr = gscale * grad + momentum * corr ;  W = (1 - l2) * W - lr * corr      (grad OxI, ld ldg)
 *   bcorr = gscale * bgrad + momentum * bcorr ;  b = b - lr * bcorr
 * Same arithmetic, element for element, as hipF_affine_update's epilogue.  corr / bcorr may be
 * NULL only when momentum == 0. */
int hipF_affine_apply(khStream_t s, int32_t O, int32_t I, const float *grad, int32_t ldg, const float *bgrad,
                      float gscale, float momentum, float lr, float l2, float *W, int32_t ldw, float *corr,
                      int32_t ldcorr, float *bias, float *bcorr);

/* The same step on `count` consecutive floats of a weight matrix's memory, taken as one long row (row padding
 * included: gradient 0 there): for a rank that steps only its shard of a layer.  corr may be NULL only when
 * momentum == 0.  Element for element the arithmetic of hipF_affine_apply. */
/* The same step on elements [first, first + count) of a parameter arena that holds several layers one after
 * another (weights and biases, row padding included: gradient 0 there), from the gradient at the same positions
 * of `grad`: element i is stepped with the hyper-parameters of the segment [begin, end) that contains it and left
 * alone when no segment does.  For a rank that steps its shard of a run of layers after a reduce-scatter.  first,
 * count and every segment bound must be multiples of 4, the three bases 16-byte aligned, at most
 * KH_MAX_SGD_SEGMENTS segments, sorted and disjoint.  corr may be NULL only when every momentum is 0.  Element
 * for element the arithmetic of hipF_affine_apply (a bias is a segment with l2 = 0). */
#define KH_MAX_SGD_SEGMENTS 32
typedef struct {
  size_t begin, end;
  float gscale, momentum, lr, l2;
} khSgdSegment;
int hipF_sgd_apply_segments(khStream_t s, int32_t nseg, const khSgdSegment *segs, size_t first, size_t count,
                            const float *grad, float *w, float *corr);
int hipF_sgd_apply_flat(khStream_t s, size_t count, const float *grad, float gscale, float momentum, float lr,
                        float l2, float *w, float *corr);

/* Xent::EvalVec on the device (nnet-loss.cc:59-94 = cu::FindRowMaxId + host compare +
 * cu::DiffXent + host sum), one pass over the posteriors:
 *   diff = post - onehot(tgt)           (diff may alias post)
 *   log_post[r] = log(max(post[r,tgt[r]], 1e-20))        (may be NULL)
 *   max_id[r]   = argmax_c post[r,c], lowest index on ties (may be NULL)
 *   counters[0] += #rows with max_id == tgt   (int64)
 *   counters[1] += #rows                      (int64)
 *   loss[0]     -= sum_r log_post[r]          (double, summed in row order)
 * counters / loss live on the device and are read back by the caller when it reports. */
int hipF_xent_eval(khStream_t s, const float *post, const int32_t *tgt, float *diff, float *log_post,
                   int32_t *max_id, long long *counters, double *loss, MatrixDim d);

/* hipF_xent_eval for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same kernels and results as
 * hipF_xent_eval; the arrival counter and the per-row scratch (used when log_post / max_id are NULL) live
 * in a __device__ array of the library (256 B + 8 B x 65536 rows per GPU, held until the process ends)
 * instead of allocated workspace.  Calls no hipMalloc / hipFree and does not initialise the library:
 * returns KH_ERR_NOT_INITIALISED unless hipK_init has run, KH_ERR_TOO_LARGE for d.rows > 65536.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams) race. */
int hipF_xent_eval_static(khStream_t s, const float *post, const int32_t *tgt, float *diff, float *log_post,
                          int32_t *max_id, long long *counters, double *loss, MatrixDim d);

/* The output layer of a training step in one call: hipF_affine_forward with KH_ACT_SOFTMAX followed by
 * hipF_xent_eval (nnet-biasedlinearity.h:58-63 + nnet-activation.h:233-236 + nnet-loss.cc:59-94), with the
 * posteriors kept in registers between the two.  Same results as the two calls.  post is always a valid N x O
 * buffer (ldpost == lddiff); with write_post == 0 its contents are unspecified afterwards (the trainer only
 * needs diff).  log_post / max_id may be NULL. */
int hipF_affine_forward_xent(khStream_t s, int trans_w, int32_t N, int32_t O, int32_t I, const float *in,
                             int32_t ldin, const float *W, int32_t ldw, const float *bias, float *post,
                             int32_t ldpost, int write_post, const int32_t *tgt, float *diff, int32_t lddiff,
                             float *log_post, int32_t *max_id, long long *counters, double *loss);

/* MseProgress::Eval / Mse::Eval (nnet-loss.cc:123-194): diff = a - b (may be NULL);
 * loss[0] += 0.5 * sum (a-b)^2 (double on device); per-row sums in row_sq (may be NULL). */
int hipF_mse_eval(khStream_t s, const float *a, const float *b, float *diff, float *row_sq, double *loss,
                  MatrixDim d);

/* Rbm::RbmUpdate (nnet-rbm.h:178-223):
 *   dW = mom*dW - (lr/N) neg_hid^T neg_vis + (lr/N) pos_hid^T pos_vis - lr*l2*W ;  W += dW
 *   da = mom*da - (lr/N) colsum(neg_vis) + (lr/N) colsum(pos_vis) ;              a += da
 *   db likewise with hid. */
int hipF_rbm_update(khStream_t s, int32_t N, int32_t H, int32_t V, const float *pos_vis, int32_t ldpv,
                    const float *pos_hid, int32_t ldph, const float *neg_vis, int32_t ldnv,
                    const float *neg_hid, int32_t ldnh, float lr, float momentum, float l2, float *W,
                    int32_t ldw, float *dW, int32_t lddw, float *vis_bias, float *dvis_bias,
                    float *hid_bias, float *dhid_bias);
/* The two halves of hipF_rbm_update, for callers that queue them on different streams (the bias half only
 * reads the four activation matrices, so it can run beside the two statistic GEMMs of the weight half). */
int hipF_rbm_update_weights(khStream_t s, int32_t N, int32_t H, int32_t V, const float *pos_vis, int32_t ldpv,
                            const float *pos_hid, int32_t ldph, const float *neg_vis, int32_t ldnv,
                            const float *neg_hid, int32_t ldnh, float lr, float momentum, float l2, float *W,
                            int32_t ldw, float *dW, int32_t lddw);
int hipF_rbm_update_bias(khStream_t s, int32_t N, int32_t H, int32_t V, const float *pos_vis, int32_t ldpv,
                         const float *pos_hid, int32_t ldph, const float *neg_vis, int32_t ldnv,
                         const float *neg_hid, int32_t ldnh, float lr, float momentum, float *vis_bias,
                         float *dvis_bias, float *hid_bias, float *dhid_bias);

/* hipF_rbm_update and the reconstruction error of the same step in one call -- the CD-1 loop's
 * mse.Eval(neg_vis, pos_vis, ...) followed, after the second propagation, by rbm.RbmUpdate(...)
 * (nnetbin/rbm-train-cd1-frmshuff.cc:139-150, nnet-loss.cc:160-194): additionally
 *   mse_loss[0] += 0.5 * sum((neg_vis - pos_vis)^2)     (double on the device, as hipF_mse_eval; no diff matrix)
 * The error only reads the two visible matrices, so where the weight launch carries the bias statistics it carries
 * the error rows too (same bits as hipF_mse_eval); otherwise the error runs as hipF_mse_eval.  ldnv must equal ldpv. */
int hipF_rbm_update_mse(khStream_t s, int32_t N, int32_t H, int32_t V, const float *pos_vis, int32_t ldpv,
                        const float *pos_hid, int32_t ldph, const float *neg_vis, int32_t ldnv,
                        const float *neg_hid, int32_t ldnh, float lr, float momentum, float l2, float *W,
                        int32_t ldw, float *dW, int32_t lddw, float *vis_bias, float *dvis_bias,
                        float *hid_bias, float *dhid_bias, double *mse_loss);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU (no counterpart in the reference, which is single-GPU: cu-device.cc).
 * One process per GPU; RCCL over xGMI.  The unique id is created on rank 0 and handed to the
 * other ranks by the caller (file, socket, torch.distributed store ...).
 * ------------------------------------------------------------------------------------------ */
#define KH_COMM_ID_BYTES 128
int hipK_comm_get_unique_id(unsigned char id[KH_COMM_ID_BYTES]);
int hipK_comm_init(const unsigned char id[KH_COMM_ID_BYTES], int rank, int nranks);
/* Single-process stand-in for `nranks` replicas that all hold exactly this process's data: a sum
 * all-reduce multiplies by nranks, min / max and the barrier are local.  Lets the data-parallel
 * code path (gradient -> all-reduce -> step) be validated on one GPU: training on a bunch X with
 * a loopback of n equals single-process training on n stacked copies of X. */
int hipK_comm_init_loopback(int nranks);
/* The stand-in as ONE rank of nranks (rank 0) for timing runs: reduce-scatter sums, and hipK_comm_shard_range
 * names, only that rank's shard, the all-gather brings nothing in -- the device work of one real rank without the
 * link traffic.  The other shards of the weights are then never stepped: the numbers of such a run mean nothing. */
int hipK_comm_init_loopback_one_rank(int nranks);
/* Real ranks WITHOUT RCCL, for validation on a box whose ranks must share one GPU (RCCL wants a device per rank):
 * the processes exchange through the file `path`, which all of them map -- a counting barrier and one slot of
 * slot_bytes (a multiple of 256, at least 4096) per rank.  Every collective below then copies device -> slot,
 * meets the other ranks, combines the slots on the host in rank order (all ranks get the same bits; a
 * reduce-scatter gives the bits of the all-reduce) and copies back; buffers larger than a slot go in pieces.
 * Blocking and slow: a test vehicle for the multi-rank code paths with different data on every rank, not a
 * transport.  `path` must not exist before the job (a new file reads as zeros, the barrier's start state) and
 * should live in memory (/dev/shm); a rank that waits two minutes for the others fails with KH_ERR_COMM, and so
 * does every later call on that file. */
int hipK_comm_init_host(const char *path, int rank, int nranks, size_t slot_bytes);
int hipK_comm_destroy(void);
int hipK_comm_rank(int *rank, int *nranks);
/* in-place sum all-reduce of count floats on stream s (ordered with the kernels on s) */
int hipF_allreduce_sum(khStream_t s, float *buf, size_t count);
/* all-gather on stream s: every rank contributes `count` floats, recv receives nranks * count (rank r's block at
 * r * count); send may be recv + rank * count */
int hipF_allgather(khStream_t s, const float *send, float *recv, size_t count);
/* Sharded exchange for data-parallel training (no counterpart in the reference, which trains on one GPU).  A
 * buffer of nranks * shard_count floats is cut into one shard per rank, shard r at offset r * shard_count:
 *   hipF_reduce_scatter_sum  in place; afterwards this rank's shard holds the sum over ranks (the other shards
 *                            are undefined)
 *   hipK_comm_shard_range    the elements this process must step: its own shard -- every shard for the loopback
 *                            stand-in, which plays all the ranks
 *   hipF_allgather_shards    in place; every rank sends its shard, afterwards all hold the whole buffer
 * hipK_comm_group_begin/end bracket collectives that may be sent as one launch (ncclGroupStart/End). */
int hipK_comm_group_begin(void);
int hipK_comm_group_end(void);
int hipK_comm_shard_range(size_t shard_count, size_t *first, size_t *count);
int hipF_reduce_scatter_sum(khStream_t s, float *buf, size_t shard_count);
int hipF_allgather_shards(khStream_t s, float *buf, size_t shard_count);
int hipK_allreduce_sum_f64(khStream_t s, double *buf, size_t count);
int hipK_allreduce_sum_i64(khStream_t s, long long *buf, size_t count);
/* in-place all-reduce of count int64 with op 0 = sum, 1 = min, 2 = max (control-plane scalars:
 * "how many bunches does every rank have", "has any rank run out of data") */
int hipK_allreduce_i64(khStream_t s, long long *buf, size_t count, int op);
int hipK_comm_barrier(khStream_t s);

#ifdef __cplusplus
}
#endif
#endif /* KALDI_AMD_INCLUDE_KALDI_HIP_H_ */
