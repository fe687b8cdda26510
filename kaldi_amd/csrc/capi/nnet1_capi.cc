#include "capi/nnet1_capi.h"
#include "nnetbin/dist-utils.h"

#include <string>

#include "cudamatrix/cu-rand.h"
#include "nnet/nnet-biasedlinearity.h"
#include "nnet/nnet-loss.h"
#include "nnet/nnet-nnet.h"
#include "nnet/nnet-rbm.h"
#include "util/table-types.h"

using namespace kaldi;

namespace {
thread_local std::string g_error;

struct Trainer {
  Nnet nnet;
  Xent xent;
  CuMatrix<BaseFloat> nnet_in, nnet_out, glob_err;
  CuStlVector<int32> targets;
  bool keep_posteriors = true;
};

struct RbmTrainer {
  Nnet nnet;
  CuRand<BaseFloat> cu_rand;
  MseProgress mse;
  CuMatrix<BaseFloat> pos_vis, pos_hid, neg_vis, neg_hid;
};

// device rows (ld) -> CuMatrix, the copy Cache::GetBunch makes
void CopyIn(const float *dev, int ld, int rows, int cols, CuMatrix<BaseFloat> *m) {
  m->Resize(rows, cols);
  CU_SAFE_CALL(hipK_memcpy_2d(CuDevice::Instantiate().Stream(), m->Data(), m->Stride() * sizeof(float), dev,
                              ld * sizeof(float), cols * sizeof(float), rows, 2));
}
}  // namespace

#define CAPI_TRY try {
#define CAPI_CATCH(ret)              \
  }                                  \
  catch (const std::exception &e) {  \
    g_error = e.what();              \
    return ret;                      \
  }

extern "C" {

const char *nnet1_last_error(void) { return g_error.c_str(); }

int nnet1_select_gpu(int gpu_id) {
  CAPI_TRY
  CuDevice::Instantiate().Verbose(false);
  CuDevice::Instantiate().SelectGpuId(gpu_id);
  return 0;
  CAPI_CATCH(-1)
}

void *nnet1_create(const char *model) {
  CAPI_TRY
  CuDevice::Instantiate().Verbose(false);
  Trainer *t = new Trainer;
  t->nnet.Read(model);
  return t;
  CAPI_CATCH(NULL)
}
void nnet1_destroy(void *h) { delete static_cast<Trainer *>(h); }
int nnet1_num_layers(void *h) { return static_cast<Trainer *>(h)->nnet.LayerCount(); }
int nnet1_input_dim(void *h) {
  CAPI_TRY
  return static_cast<Trainer *>(h)->nnet.InputDim();
  CAPI_CATCH(-1)
}
int nnet1_output_dim(void *h) {
  CAPI_TRY
  return static_cast<Trainer *>(h)->nnet.OutputDim();
  CAPI_CATCH(-1)
}

int nnet1_set_train_opts(void *h, float learn_rate, float momentum, float l2_penalty, float l1_penalty,
                         float l2_upper_bound, int average_grad, int fused, const char *learn_factors) {
  CAPI_TRY
  Nnet &n = static_cast<Trainer *>(h)->nnet;
  n.SetLearnRate(learn_rate, (learn_factors && *learn_factors) ? learn_factors : NULL);
  n.SetConstantMomentum(momentum);
  n.SetL2Penalty(l2_penalty);
  n.SetL2UpperBound(l2_upper_bound);
  n.SetL1Penalty(l1_penalty);
  n.SetAverageGrad(average_grad != 0);
  n.SetFusion(fused != 0);
  return 0;
  CAPI_CATCH(-1)
}

int nnet1_set_data_parallel(void *h, int nranks) {
  CAPI_TRY
  static_cast<Trainer *>(h)->nnet.SetDataParallel(nranks);
  return 0;
  CAPI_CATCH(-1)
}

int nnet1_train_bunch(void *h, const float *feats_dev, int ld, int rows, int cols, const int32_t *labels_dev,
                      int cross_validate) {
  CAPI_TRY
  Trainer *t = static_cast<Trainer *>(h);
  // windows onto the caller's rows and labels, the way the tool takes bunches out of its cache
  // (Cache::GetBunchView); the caller keeps them unchanged until this call's work has finished
  t->nnet_in.AttachView(const_cast<float *>(feats_dev), rows, cols, ld);
  t->targets.AttachView(const_cast<int32 *>(labels_dev), rows);
  t->nnet.SetBorrowInput(true);
  t->nnet.PropagateXent(t->nnet_in, t->targets, &t->xent, &t->nnet_out, &t->glob_err, t->keep_posteriors);
  if (!cross_validate) t->nnet.Backpropagate(t->glob_err, NULL);
  return 0;
  CAPI_CATCH(-1)
}

int nnet1_keep_posteriors(void *h, int keep) {
  CAPI_TRY
  static_cast<Trainer *>(h)->keep_posteriors = keep != 0;
  return 0;
  CAPI_CATCH(-1)
}

int nnet1_fusion_enabled(void *h) {
  CAPI_TRY
  return static_cast<Trainer *>(h)->nnet.GetFusion() ? 1 : 0;
  CAPI_CATCH(-1)
}

int nnet1_feedforward(void *h, const float *feats_dev, int ld, int rows, int cols, float *out_dev, int ld_out) {
  CAPI_TRY
  Trainer *t = static_cast<Trainer *>(h);
  CopyIn(feats_dev, ld, rows, cols, &t->nnet_in);
  t->nnet.Feedforward(t->nnet_in, &t->nnet_out);
  CU_SAFE_CALL(hipK_memcpy_2d(CuDevice::Instantiate().Stream(), out_dev, ld_out * sizeof(float), t->nnet_out.Data(),
                              t->nnet_out.Stride() * sizeof(float), t->nnet_out.NumCols() * sizeof(float),
                              t->nnet_out.NumRows(), 2));
  return 0;
  CAPI_CATCH(-1)
}

int nnet1_get_stats(void *h, long long *frames, long long *correct, double *loss) {
  CAPI_TRY
  int64 f, c;
  double l;
  static_cast<Trainer *>(h)->xent.GetStats(&f, &c, &l);
  if (frames) *frames = f;
  if (correct) *correct = c;
  if (loss) *loss = l;
  return 0;
  CAPI_CATCH(-1)
}
int nnet1_reset_stats(void *h) {
  CAPI_TRY
  static_cast<Trainer *>(h)->xent.Reset();
  return 0;
  CAPI_CATCH(-1)
}
int nnet1_write(void *h, const char *wx, int binary) {
  CAPI_TRY
  static_cast<Trainer *>(h)->nnet.Write(wx, binary != 0);
  return 0;
  CAPI_CATCH(-1)
}
int nnet1_get_posteriors(void *h, float *host, int rows, int cols) {
  CAPI_TRY
  Trainer *t = static_cast<Trainer *>(h);
  // after EvalVec the posteriors live in nnet_out (glob_err holds posterior - target)
  if (t->nnet_out.NumRows() != rows || t->nnet_out.NumCols() != cols) KALDI_ERR << "shape mismatch";
  Matrix<BaseFloat> m;
  t->nnet_out.CopyToMat(&m);
  memcpy(host, m.Data(), sizeof(float) * rows * cols);
  return 0;
  CAPI_CATCH(-1)
}
int nnet1_get_layer_params(void *h, int layer, float *w, float *b) {
  CAPI_TRY
  Trainer *t = static_cast<Trainer *>(h);
  if (layer < 0 || layer >= t->nnet.LayerCount() || t->nnet.Layer(layer)->GetType() != Component::kBiasedLinearity)
    KALDI_ERR << "layer " << layer << " is not a <biasedlinearity>";
  BiasedLinearity *bl = static_cast<BiasedLinearity *>(t->nnet.Layer(layer));
  Matrix<BaseFloat> m;
  bl->GetLinearityWeight().CopyToMat(&m);
  memcpy(w, m.Data(), sizeof(float) * m.NumRows() * m.NumCols());
  Vector<BaseFloat> v;
  bl->GetBiasWeight().CopyToVec(&v);
  memcpy(b, v.Data(), sizeof(float) * v.Dim());
  return 0;
  CAPI_CATCH(-1)
}

void *rbm1_create(const char *model) {
  CAPI_TRY
  CuDevice::Instantiate().Verbose(false);
  RbmTrainer *t = new RbmTrainer;
  t->nnet.Read(model);
  if (t->nnet.LayerCount() != 1 || t->nnet.Layer(0)->GetType() != Component::kRbm) {
    delete t;
    KALDI_ERR << "rbm1_create: the model must be exactly one <rbm>";
  }
  return t;
  CAPI_CATCH(NULL)
}
void rbm1_destroy(void *h) { delete static_cast<RbmTrainer *>(h); }
int rbm1_set_train_opts(void *h, float learn_rate, float momentum, float l2_penalty) {
  CAPI_TRY
  RbmBase &rbm = dynamic_cast<RbmBase &>(*static_cast<RbmTrainer *>(h)->nnet.Layer(0));
  rbm.SetLearnRate(learn_rate);
  rbm.SetMomentum(momentum);
  rbm.SetL2Penalty(l2_penalty);
  return 0;
  CAPI_CATCH(-1)
}
int rbm1_cd1_bunch(void *h, const float *feats_dev, int ld, int rows, int cols) {
  CAPI_TRY
  RbmTrainer *t = static_cast<RbmTrainer *>(h);
  RbmBase &rbm = dynamic_cast<RbmBase &>(*t->nnet.Layer(0));
  // a window onto the caller's rows (read-only here), like a bunch taken out of the cache
  t->pos_vis.AttachView(const_cast<float *>(feats_dev), rows, cols, ld);
  if (rbm.HidType() == Rbm::BERNOULLI) {
    rbm.PropagateSample(t->pos_vis, &t->cu_rand, &t->pos_hid, &t->neg_hid);
  } else {
    rbm.Propagate(t->pos_vis, &t->pos_hid);
    t->neg_hid.CopyFromMat(t->pos_hid);
    t->cu_rand.AddGaussNoise(&t->neg_hid);
  }
  rbm.Reconstruct(t->neg_hid, &t->neg_vis);
  rbm.Propagate(t->neg_vis, &t->neg_hid);
  rbm.RbmUpdate(t->pos_vis, t->pos_hid, t->neg_vis, t->neg_hid, t->mse.DeviceLoss());  // (with the reconstruction error)
  t->mse.Evaluated(t->neg_vis.NumRows());
  return 0;
  CAPI_CATCH(-1)
}
int rbm1_write(void *h, const char *wx, int binary) {
  CAPI_TRY
  static_cast<RbmTrainer *>(h)->nnet.Write(wx, binary != 0);
  return 0;
  CAPI_CATCH(-1)
}
int rbm1_dims(void *h, int *vis, int *hid) {
  CAPI_TRY
  Nnet &n = static_cast<RbmTrainer *>(h)->nnet;
  if (vis) *vis = n.InputDim();
  if (hid) *hid = n.OutputDim();
  return 0;
  CAPI_CATCH(-1)
}
int nnet1_rand_stream(unsigned int seed, int n, int *out) {
  SRand(seed);
  for (int i = 0; i < n; ++i) out[i] = Rand();
  SRand(1);
  return 0;
}
int nnet1_synchronize(void) {
  CAPI_TRY
  CuDevice::Instantiate().Synchronize();
  return 0;
  CAPI_CATCH(-1)
}


int nnet1_shard_assign(int num_ranks, int n, const long long *num_frames, int *rank_out) {
  if (num_ranks < 1 || n < 0 || (n > 0 && (!num_frames || !rank_out))) return -1;
  kaldi::ShardAssigner shard(num_ranks);
  for (int i = 0; i < n; ++i) rank_out[i] = shard.Assign(num_frames[i]);
  return 0;
}

int nnet1_substitute_job(const char *spec, int rank, char *out, int out_len) {
  if (!spec || !out || out_len <= 0 || rank < 0) return -1;
  const std::string r = kaldi::SubstituteJob(spec, rank);
  if (static_cast<int>(r.size()) + 1 > out_len) return -1;
  memcpy(out, r.c_str(), r.size() + 1);
  return 0;
}

int nnet1_table_scan_shard(const char *rspecifier, int num_ranks, int rank, long long *kept, long long *kept_frames,
                           long long *payload_bytes) {
  CAPI_TRY
  if (!rspecifier || num_ranks < 1 || rank < 0 || rank >= num_ranks) return -1;
  kaldi::ShardAssigner shard(num_ranks);
  kaldi::SequentialBaseFloatMatrixReader reader(
      rspecifier, [&shard, rank](const std::string &, int32 rows) { return shard.Assign(rows) == rank; });
  long long n = 0, frames = 0;
  for (; !reader.Done(); reader.Next()) {
    ++n;
    frames += reader.Value().NumRows();
  }
  if (kept) *kept = n;
  if (kept_frames) *kept_frames = frames;
  if (payload_bytes) *payload_bytes = static_cast<long long>(reader.PayloadBytesRead());
  return 0;
  CAPI_CATCH(-1)
}

int nnet1_comm_nonce(long parent_pid, char *out, int out_len) {
  if (!out || out_len <= 0) return -1;
  const std::string r = kaldi::CommNonce(parent_pid);
  if (static_cast<int>(r.size()) + 1 > out_len) return -1;
  memcpy(out, r.c_str(), r.size() + 1);
  return 0;
}

int nnet1_comm_id_file_write(const char *file, const char *nonce, const unsigned char *id, int id_bytes) {
  if (!file || !nonce || !id || id_bytes <= 0) return -1;
  return kaldi::WriteCommIdFile(file, nonce, id, id_bytes) ? 0 : -1;
}

int nnet1_comm_id_file_read(const char *file, const char *nonce, unsigned char *id, int id_bytes) {
  if (!file || !nonce || !id || id_bytes <= 0) return -1;
  return kaldi::ReadCommIdFile(file, nonce, id, id_bytes) ? 1 : 0;
}

}  // extern "C"
