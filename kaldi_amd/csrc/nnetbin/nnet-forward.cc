// nnet-forward: forward pass of a network over a feature table.  Command line, options, NaN/inf
// checks, log lines and exit status of the reference tool (nnetbin/nnet-forward.cc:27-182); with
// --num-ranks > 1 the utterances are sharded over the ranks (balanced by frames, no communication)
// and "JOB" in the wspecifier is replaced by rank+1.
//
// The reference handles one utterance at a time from the read to the write.  Here three threads and two
// copy streams work on different utterances at once: a reader thread reads ahead, this thread uploads an
// utterance and queues its kernels, and the outputs travel to the host and into the table behind it
// (PosteriorWriter).  Every utterance is still computed on its own, by the same kernels as before, so the
// values written do not depend on what else is in flight.
//
// Command line, option names, usage text and log lines are derived from nnetbin/nnet-forward.cc of troylee/kaldi
// (Copyright 2011 Karel Vesely), licensed under the Apache License, Version 2.0
// (http://www.apache.org/licenses/LICENSE-2.0); this file is distributed under the same license, "AS IS",
// without warranties or conditions of any kind.
#include <limits>

#include "base/kaldi-common.h"
#include "cudamatrix/cu-device.h"
#include "nnet/nnet-nnet.h"
#include "nnetbin/dist-utils.h"
#include "nnetbin/feature-upload.h"
#include "nnetbin/posterior-download.h"
#include "util/common-utils.h"
#include "util/timer.h"

int main(int argc, char *argv[]) {
  using namespace kaldi;
  try {
    const char *usage =
        "Perform forward pass through Neural Network.\n"
        "Usage:  nnet-forward [options] <model-in> <feature-rspecifier> <feature-wspecifier>\n"
        "e.g.: \n"
        " nnet-forward nnet ark:features.ark ark:mlpoutput.ark\n";

    ParseOptions po(usage);
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform Neural Network");
    std::string class_frame_counts;
    po.Register("class-frame-counts", &class_frame_counts, "Counts of frames for posterior division by class-priors");
    BaseFloat prior_scale = 1.0;
    po.Register("prior-scale", &prior_scale, "scaling factor of prior log-probabilites given by --class-frame-counts");
    bool apply_log = false, silent = false, no_softmax = false;
    po.Register("apply-log", &apply_log, "Transform MLP output to logscale");
    po.Register("no-softmax", &no_softmax,
                "No softmax on MLP output. The MLP outputs directly log-likelihoods, log-priors will be subtracted");
    po.Register("silent", &silent, "Don't print any messages");
    bool fused = true;
    po.Register("fused", &fused, "Run the linearity+activation pairs as fused GPU kernels");
    int32 batch_frames = 4096;
    po.Register("batch-frames", &batch_frames,
                "Run the network on several utterances at a time, about this many frames per launch chain "
                "(0: one utterance at a time).  The outputs do not depend on it.");
    DistOptions dist;
    dist.Register(&po);

    po.Read(argc, argv);
    if (po.NumArgs() != 3) {
      po.PrintUsage();
      exit(1);
    }
    std::string model_filename = po.GetArg(1), feature_rspecifier = po.GetArg(2), feature_wspecifier = po.GetArg(3);

    dist.Resolve();
    if (silent) CuDevice::Instantiate().Verbose(false);
    DistInit(dist, /*need_comm=*/false);  // shards are independent: no collective on this path
    if (dist.num_ranks > 1) feature_wspecifier = SubstituteJob(feature_wspecifier, dist.rank);

    Nnet nnet_transf;
    if (feature_transform != "") nnet_transf.Read(feature_transform);
    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetFusion(fused);
    nnet_transf.SetFusion(fused);

    kaldi::int64 tot_t = 0;
    // Every rank walks the same table and keeps the utterances the assigner gives it (the rank with the fewest
    // frames so far), decided on each entry's header: a binary matrix of another rank is skipped unread.  The
    // reader reads ahead on a thread of its own; the assigner is only used there.
    ShardAssigner shard(dist.num_ranks);
    const int32 my_rank = dist.rank;
    kaldi::int64 table_entries = 0;  // utterances of the whole table (written by the reader's thread, read after the loop)
    PrefetchingBaseFloatMatrixReader feature_reader(
        feature_rspecifier, [&shard, &table_entries, my_rank](const std::string &, int32 rows) {
          ++table_entries;
          return shard.Assign(rows) == my_rank;
        });
    // outputs are scanned for NaN / inf on the GPU, copied and written behind this thread
    PosteriorWriter feature_writer(feature_wspecifier);
    FeatureUploader uploader(8192);

    CuMatrix<BaseFloat> feats_transf;

    // class priors from frame counts: log-domain offset or linear-domain scale
    Vector<BaseFloat> tmp_priors;
    CuVector<BaseFloat> priors;
    if (class_frame_counts != "") {
      Input in;
      in.OpenTextMode(class_frame_counts);
      tmp_priors.Read(in.Stream(), false);
      in.Close();
      tmp_priors.Scale(1.0 / tmp_priors.Sum());
      if (apply_log || no_softmax) {
        tmp_priors.ApplyLog();
        tmp_priors.Scale(-prior_scale);
      } else {
        tmp_priors.ApplyPow(-prior_scale);
      }
      priors.CopyFromVec(tmp_priors);
    }

    Timer tim;
    if (!silent) KALDI_LOG << "MLP FEEDFORWARD STARTED";

    // Several utterances share one pass through the network: their (transformed) rows are gathered in one device
    // matrix and the launch chain runs once over all of them -- a 9000-state Switchboard net does 1.34 M frames/s
    // on 1024-4096 rows at a time against 1.10 M utterance by utterance (300-1200 rows).  The feature transform
    // still runs per utterance (a splice must not reach into a neighbour).  With batching on, every product sums a
    // row's K range in one piece (hipK_gemm_row_invariant), so an utterance's output has the same bits whatever
    // shares its batch: one process and several ranks, which cut the table into different batches, write the same.
    // Only for networks whose every component treats the rows of its input one by one: a splice (<expand>) in the
    // model itself reaches into neighbouring frames, and <dropout> draws from a generator whose state is laid out
    // by matrix size.  Such models run one utterance at a time, as before, with the same row-invariant products.
    bool rows_independent = true;
    for (int32 i = 0; i < nnet.LayerCount(); ++i) {
      const Component::ComponentType t = nnet.Layer(i)->GetType();
      if (t != Component::kBiasedLinearity && t != Component::kSoftmax && t != Component::kSigmoid && t != Component::kRelu &&
          t != Component::kSoftRelu && t != Component::kRbm && t != Component::kCopy)
        rows_independent = false;
    }
    const bool batching = batch_frames > 1 && rows_independent;
    // (also without batching: --batch-frames=0 and models that run one utterance at a time write the same bits)
    CU_SAFE_CALL(hipK_gemm_row_invariant(1));
    CuMatrix<BaseFloat> batch_in, batch_view;
    int32 batch_rows = 0;
    auto finish_output = [&](CuMatrix<BaseFloat> *nnet_out) {
      if (apply_log) nnet_out->ApplyLog();
      if (class_frame_counts != "") {
        if (apply_log || no_softmax) nnet_out->AddVecToRows(1.0, priors, 1.0);
        else nnet_out->MulColsVec(priors);
      }
    };
    auto flush = [&]() {
      if (!feature_writer.BatchOpen()) return;
      const int32 out_dim = nnet.LayerCount() > 0 ? nnet.OutputDim() : batch_in.NumCols();
      CuMatrix<BaseFloat> *nnet_out = feature_writer.Finish(batch_rows, out_dim);
      if (batch_rows > 0) {
        batch_view.AttachRows(batch_in, 0, batch_rows);
        if (nnet.LayerCount() > 0) nnet.Feedforward(batch_view, nnet_out);
        else nnet_out->AssignInPlace(batch_view);
        finish_output(nnet_out);
      }
      feature_writer.SubmitBatch();
      batch_rows = 0;
    };

    int32 num_done = 0;
    for (; !feature_reader.Done(); feature_reader.Next()) {
      const Matrix<BaseFloat> &mat = feature_reader.Value();
      const CuMatrix<BaseFloat> &feats = uploader.Upload(mat);
      nnet_transf.Feedforward(feats, &feats_transf);
      if (!batching) {
        CuMatrix<BaseFloat> *nnet_out = feature_writer.Begin(mat.NumRows(), nnet.LayerCount() > 0 ? nnet.OutputDim() : feats_transf.NumCols());
        if (nnet.LayerCount() > 0) nnet.Feedforward(feats_transf, nnet_out);
        else if (nnet_out->NumRows() > 0) nnet_out->AssignInPlace(feats_transf);
        finish_output(nnet_out);
        feature_writer.Submit(feature_reader.Key(), feats);
      } else {
        const int32 rows = feats_transf.NumRows(), cols = feats_transf.NumCols();
        // (an empty utterance has no columns either: it rides in whatever batch is open)
        if (feature_writer.BatchOpen() &&
            (feature_writer.BatchPieces() == PosteriorWriter::kMaxPieces ||
             (rows > 0 && (batch_rows + rows > batch_in.NumRows() || cols != batch_in.NumCols()))))
          flush();
        if (rows > 0 && (batch_in.NumCols() != cols || batch_in.NumRows() < rows)) {
          KALDI_ASSERT(batch_rows == 0);
          batch_in.Resize(std::max(batch_frames, rows), cols);
        }
        if (!feature_writer.BatchOpen()) feature_writer.BeginBatch();
        feature_writer.AddPiece(feature_reader.Key(), feats, rows);
        if (rows > 0) batch_in.CopyRowsFromMat(rows, feats_transf, 0, batch_rows);
        batch_rows += rows;
        if (batch_rows >= batch_frames) flush();
      }
      if (num_done % 1000 == 0 && !silent) KALDI_LOG << num_done << ", " << std::flush;
      num_done++;
      tot_t += mat.NumRows();
    }
    flush();

    if (!feature_writer.Close()) KALDI_ERR << "Error closing the output table " << feature_wspecifier;
    if (!silent) KALDI_LOG << "MLP FEEDFORWARD FINISHED " << tim.Elapsed() << "s, fps" << tot_t / tim.Elapsed();
    if (!silent) KALDI_LOG << "Done " << num_done << " files";
    if (!silent && dist.num_ranks > 1)
      KALDI_LOG << "rank " << dist.rank << " of " << dist.num_ranks << ": feature payload read "
                << feature_reader.PayloadBytesRead() << " bytes";
    if (!silent) CuDevice::Instantiate().PrintProfile();
    DistFinalize(dist, /*need_comm=*/false);
    // Like the reference, a run that found nothing to do fails (nnet-forward.cc:177).  A rank of a sharded job whose
    // table has fewer utterances than there are ranks has nothing to do although the job has: it succeeds, and leaves
    // an empty archive behind.
    return ((num_done > 0 || (dist.num_ranks > 1 && table_entries > 0)) ? 0 : 1);
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
