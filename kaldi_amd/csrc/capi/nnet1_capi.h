/* C entry points over the host C++ layer (Nnet, Xent, Rbm, CuRand), for callers that are not C++:
 * the benchmark and the Python tests drive exactly the objects the nnetbin tools use.
 * Handles are opaque; every function returns 0 on success and -1 on error (text via
 * nnet1_last_error()).  Device pointers are plain float / int32 device pointers. */
#ifndef KALDI_AMD_CAPI_NNET1_CAPI_H_
#define KALDI_AMD_CAPI_NNET1_CAPI_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
const char *nnet1_last_error(void);
int nnet1_select_gpu(int gpu_id);
/* network + cross-entropy trainer */
void *nnet1_create(const char *model_rxfilename);
void nnet1_destroy(void *h);
int nnet1_num_layers(void *h);
int nnet1_input_dim(void *h);
int nnet1_output_dim(void *h);
/* the options of one run of the trainer tool, whose momentum is one value per process: with momentum 0 the
 * layers do not keep the last gradient, so a later call that turns momentum on starts from a zero correction
 * (Nnet::SetMomentum on the C++ side keeps it, as the reference does) */
int nnet1_set_train_opts(void *h, float learn_rate, float momentum, float l2_penalty, float l1_penalty,
                         float l2_upper_bound, int average_grad, int fused, const char *learn_factors);
int nnet1_set_data_parallel(void *h, int nranks);
/* one minibatch of nnet-train-xent-hardlab-frmshuff's inner loop: copy the bunch out of a
 * (cache) buffer, Propagate, Xent::EvalVec, Backpropagate (skipped when cross_validate != 0) */
int nnet1_train_bunch(void *h, const float *feats_dev, int ld, int rows, int cols, const int32_t *labels_dev,
                      int cross_validate);
/* keep != 0 (default): the posteriors of the last bunch stay readable (nnet1_get_posteriors);
 * 0: the training step never writes them (the output layer, its softmax and the loss are fused) */
int nnet1_keep_posteriors(void *h, int keep);
/* 1 while the network runs its fused passes (Nnet::SetFusion; a look at Nnet::PropagateBuffer() turns them off) */
int nnet1_fusion_enabled(void *h);
/* Nnet::Feedforward; out_dev is rows x output_dim with leading dimension ld_out */
int nnet1_feedforward(void *h, const float *feats_dev, int ld, int rows, int cols, float *out_dev, int ld_out);
int nnet1_get_stats(void *h, long long *frames, long long *correct, double *loss);
int nnet1_reset_stats(void *h);
int nnet1_write(void *h, const char *wxfilename, int binary);
/* copies of internal tensors for parity checks (host buffers, dense row-major) */
int nnet1_get_posteriors(void *h, float *host, int rows, int cols);
int nnet1_get_layer_params(void *h, int layer, float *weights_host, float *bias_host);
/* RBM CD-1 trainer */
void *rbm1_create(const char *model_rxfilename);
void rbm1_destroy(void *h);
int rbm1_set_train_opts(void *h, float learn_rate, float momentum, float l2_penalty);
/* one minibatch of rbm-train-cd1-frmshuff's inner loop */
int rbm1_cd1_bunch(void *h, const float *feats_dev, int ld, int rows, int cols);
int rbm1_write(void *h, const char *wxfilename, int binary);
int rbm1_dims(void *h, int *vis, int *hid);
int nnet1_synchronize(void);
/* the host random stream (kaldi::Rand after kaldi::SRand(seed)): n draws, for tests */
int nnet1_rand_stream(unsigned int seed, int n, int *out);
/* Host-only helpers of the multi-GPU tools (no GPU needed): the utterance -> rank rule of the
 * sharded nnet-forward (each utterance to the rank with the fewest frames so far) and the JOB
 * substitution in output specifiers. */
int nnet1_shard_assign(int num_ranks, int n, const long long *num_frames, int *rank_out);
int nnet1_substitute_job(const char *spec, int rank, char *out, int out_len);
/* Walks a table of float matrices as rank `rank` of the sharded nnet-forward does (other ranks' entries are
 * skipped on their headers): entries and frames kept, payload bytes brought in.  -1 on any read error, a
 * table cut off inside a skipped entry included. */
int nnet1_table_scan_shard(const char *rspecifier, int num_ranks, int rank, long long *kept, long long *kept_frames,
                           long long *payload_bytes);
/* The launch nonce of the RCCL id file as a rank whose parent process is parent_pid computes it from the
 * environment, and the id file's writer / reader (read: 1 = this launch's complete file, 0 = not yet). */
int nnet1_comm_nonce(long parent_pid, char *out, int out_len);
int nnet1_comm_id_file_write(const char *file, const char *nonce, const unsigned char *id, int id_bytes);
int nnet1_comm_id_file_read(const char *file, const char *nonce, unsigned char *id, int id_bytes);
#ifdef __cplusplus
}
#endif
#endif
