#!/usr/bin/env python3
"""Benchmark of the nnet1 training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1; N>1 is launched by torch.distributed.run)

A step is one minibatch of nnet-train-xent-hardlab-frmshuff's inner loop -- the bunch copy out of
the frame cache, Nnet::Propagate, Xent::EvalVec, Nnet::Backpropagate with the SGD update --
driven through the same C++ objects the tool uses (libkaldi_nnet.so).  Workload: the
Switchboard-scale 1320 -> 7x2048 sigmoid -> 9000 softmax DNN, bunch 256, fp32, synthetic
features N(0,1), uniform labels, weights 0.1*N(0,1), hidden biases U(-4.1,-3.9), lr 0.008,
momentum 0.  With N > 1 every rank trains on its own bunches and the per-layer gradients are summed
with an RCCL all-reduce (synchronous data parallel, weak scaling: 256 frames per GPU per step).

Prints ONE JSON line on rank 0: frames/s over all GPUs, the roofline of the dominant kernel (the
fp32 MFMA GEMM: algorithmic FLOPs over the time of its launches, HIP events on their stream) and
the reference CPU implementation timed on this box's host cores on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz)
WORKLOADS = {
    # kind "train": one bunch of nnet-train-xent-hardlab-frmshuff; "forward": one batch of nnet-forward
    # (Nnet::Feedforward on utterances of 300..1200 frames gathered to >= 4096); "rbm": one bunch of rbm-train-cd1-frmshuff
    "swbd": dict(kind="train", dims=[1320] + [2048] * 7 + [9000], name="swbd-1320-7x2048-9000-xent-train"),
    "rm": dict(kind="train", dims=[440, 1024, 1024, 1024, 1024, 1483], name="rm-440-4x1024-1483-xent-train"),
    "swbd-forward": dict(kind="forward", dims=[1320] + [2048] * 7 + [9000], name="swbd-1320-7x2048-9000-forward"),
    "rm-forward": dict(kind="forward", dims=[440, 1024, 1024, 1024, 1024, 1483], name="rm-440-4x1024-1483-forward"),
    "rbm-gauss-bern": dict(kind="rbm", dims=[440, 2048], types=("gauss", "bern"), lr=0.001, mom=0.5, l2=2e-4,
                           name="rbm-440x2048-gauss-bern-cd1"),
    "rbm-bern-bern": dict(kind="rbm", dims=[2048, 2048], types=("bern", "bern"), lr=0.1, mom=0.9, l2=2e-4,
                          name="rbm-2048x2048-bern-bern-cd1"),
}
METRICS = {"train": "DNN train frames/sec (fwd+bwd+update)", "forward": "DNN forward frames/sec (nnet-forward)",
           "rbm": "RBM CD-1 frames/sec"}
BUNCH = int(os.environ.get("KALDI_BENCH_BUNCH", "256"))  # (the variable: probes at other bunch sizes; the benchmark is 256)


def flops_per_frame(dims, kind="train"):
    """2*MAC over the GEMMs of one frame.  Training: forward 2P, weight gradient 2P, input gradient
    2(P - P1) (the first layer's input gradient is never computed).  Forward: 2P.  RBM CD-1: five
    V x H products (three propagations / reconstructions, two statistics)."""
    p = [dims[i] * dims[i + 1] for i in range(len(dims) - 1)]
    if kind == "forward":
        return 2 * sum(p)
    if kind == "rbm":
        return 10 * p[0]
    return 6 * sum(p) - 2 * p[0]


def write_model(path, dims, seed):
    """random-init sigmoid DNN in the nnet1 binary format"""
    rs = np.random.RandomState(seed)
    with open(path, "wb") as f:
        f.write(b"\0B")
        for i in range(len(dims) - 1):
            o, n = dims[i + 1], dims[i]
            last = i == len(dims) - 2
            W = (0.1 * rs.standard_normal((o, n))).astype("<f4")
            b = np.zeros(o, "<f4") if last else (-4.1 + 0.2 * rs.random_sample(o)).astype("<f4")
            hdr = lambda tok, a, c: tok + b" " + b"\x04" + np.int32(a).tobytes() + b"\x04" + np.int32(c).tobytes()
            f.write(hdr(b"<biasedlinearity>", o, n))
            f.write(hdr(b"FM", o, n))
            f.write(W.tobytes())
            f.write(b"FV " + b"\x04" + np.int32(o).tobytes())
            f.write(b.tobytes())
            f.write(hdr(b"<softmax>" if last else b"<sigmoid>", o, o))


def write_rbm(path, vis, hid, vis_type, hid_type, seed):
    """random-init RBM in the nnet1 binary format (weights 0.1*N(0,1), biases 0: gen_rbm_init.py's scheme)"""
    rs = np.random.RandomState(seed)
    i32 = lambda v: b"\x04" + np.int32(v).tobytes()
    with open(path, "wb") as f:
        f.write(b"\0B<rbm> " + i32(hid) + i32(vis) + vis_type.encode() + b" " + hid_type.encode() + b" ")
        f.write(b"FM " + i32(hid) + i32(vis) + (0.1 * rs.standard_normal((hid, vis))).astype("<f4").tobytes())
        f.write(b"FV " + i32(vis) + np.zeros(vis, "<f4").tobytes())
        f.write(b"FV " + i32(hid) + np.zeros(hid, "<f4").tobytes())


def usable_cpus():
    """CPUs this process may run on: the scheduler affinity, capped by the cgroup's CPU quota and by
    OMP_NUM_THREADS (the GPU boxes export the share of the machine that belongs to one GPU there)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    omp = int(os.environ.get("OMP_NUM_THREADS", "0"))
    return min(n, omp) if omp > 0 else n


def _ref_tool(name):
    return os.path.join(ROOT, "oracle", "_ref", "bin", name)


def _run_ref(cmd, threads, pattern):
    """Runs one of the reference's tools with `threads` BLAS threads; -> (fps, loop seconds, wall seconds) from its
    own log line, or None."""
    env = dict(os.environ, OPENBLAS_NUM_THREADS=str(threads), OMP_NUM_THREADS=str(threads))
    t0 = time.time()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    wall = time.time() - t0
    m = re.search(pattern, r.stderr.decode())
    if r.returncode != 0 or not m:
        return None
    return float(m.group(2)), float(m.group(1)), round(wall, 1)


def _synthetic_archive(tmp, dims, frames, seed, labels=True, unit_interval=False):
    """utterances of 512 frames (features N(0,1), or U(0,1) for Bernoulli visible units; uniform labels) as archives"""
    from oracle import nnet1_oracle as O
    rs = np.random.RandomState(seed)
    nutt = max(1, frames // 512)
    draw = (lambda: rs.random_sample((512, dims[0]))) if unit_interval else (lambda: rs.standard_normal((512, dims[0])))
    utts = [("utt%03d" % i, draw().astype(np.float32)) for i in range(nutt)]
    fa = os.path.join(tmp, "cpu_feats.ark")
    open(fa, "wb").write(O.write_matrix_ark(utts))
    aa = None
    if labels:
        aa = os.path.join(tmp, "cpu_ali.ark")
        open(aa, "wb").write(O.write_int32_vector_ark([(k, rs.randint(0, dims[-1], 512)) for k, _ in utts]))
    return fa, aa, nutt * 512


def cpu_baseline(dims, model_path, tmp, frames_many=32768, frames_one=4096):
    """The reference's own nnet-train-xent-hardlab-frmshuff (built from /root/reference into oracle/_ref with
    HAVE_CUDA off, BLAS = OpenBLAS) on synthetic frames of the same workload: `frames_many` frames with one BLAS
    thread per usable CPU, and `frames_one` frames with a single thread; throughput = the tool's own fps line (its
    main loop: archive reading, cache, training; model load and write excluded;
    nnetbin/nnet-train-xent-hardlab-frmshuff.cc:190).  Falls back to the numpy oracle."""
    exe = _ref_tool("nnet-train-xent-hardlab-frmshuff")
    cores = usable_cpus()
    from oracle import nnet1_oracle as O

    def run_tool(frames, threads):
        fa, aa, n = _synthetic_archive(tmp, dims, frames, 11)
        r = _run_ref([exe, "--bunchsize=%d" % BUNCH, "--cachesize=%d" % min(n, 32768), "--randomize=false", "--binary=true",
                      "ark:" + fa, "ark:" + aa, model_path, os.path.join(tmp, "cpu_out.nnet")], threads,
                     r"TRAINING FINISHED (\S+)s, fps(\S+),")
        return None if r is None else dict(fps=r[0], loop_s=r[1], wall_s=r[2], frames=n)

    if os.path.exists(exe):
        many = run_tool(frames_many, cores)
        one = run_tool(frames_one, 1) if many and frames_one else None
        if many:
            out = dict(value=many["fps"], unit="frames/s", cores=cores, kind="reference",
                       sample="%d frames (utterances of 512), bunch %d, same model; the tool's own fps line "
                              "(main loop %.1f s of %.1f s wall)" % (many["frames"], BUNCH, many["loop_s"], many["wall_s"]),
                       blas="OpenBLAS (scipy), %d threads" % cores)
            if one:
                out["one_thread"] = dict(value=one["fps"], cores=1, sample="%d frames, main loop %.1f s" % (one["frames"], one["loop_s"]))
            return out
    # numpy restatement of the same step
    rs = np.random.RandomState(11)
    layers = O.read_nnet(open(model_path, "rb").read())
    net = O.Nnet(layers)
    net.set_train_opts(0.008)
    x = rs.standard_normal((BUNCH, dims[0])).astype(np.float32)
    t = rs.randint(0, dims[-1], BUNCH)
    t0 = time.time()
    n = 0
    while time.time() - t0 < 10.0:
        post = net.propagate(x)
        diff, _, _, _ = O.Xent().eval_vec(post, t)
        net.backpropagate(diff)
        n += BUNCH
    return dict(value=n / (time.time() - t0), unit="frames/s", cores=cores, kind="port",
                sample="numpy oracle, %d frames" % n)


def cpu_baseline_forward(dims, model_path, tmp, frames=16384):
    """The reference's nnet-forward (CPU build) on `frames` synthetic frames of the same model, posteriors to
    /dev/null, one BLAS thread per usable CPU: its own fps line (nnetbin/nnet-forward.cc:170: the loop over
    utterances, reading and writing included).  Falls back to the numpy oracle."""
    exe = _ref_tool("nnet-forward")
    cores = usable_cpus()
    if os.path.exists(exe):
        fa, _, n = _synthetic_archive(tmp, dims, frames, 12, labels=False)
        r = _run_ref([exe, model_path, "ark:" + fa, "ark:/dev/null"], cores, r"MLP FEEDFORWARD FINISHED (\S+)s, fps(\S+)")
        if r:
            return dict(value=r[0], unit="frames/s", cores=cores, kind="reference",
                        sample="%d frames (utterances of 512), same model, output to /dev/null; the tool's own fps line "
                               "(loop %.1f s of %.1f s wall)" % (n, r[1], r[2]), blas="OpenBLAS (scipy), %d threads" % cores)
    from oracle import nnet1_oracle as O
    net = O.Nnet(O.read_nnet(open(model_path, "rb").read()))
    x = np.random.RandomState(12).standard_normal((512, dims[0])).astype(np.float32)
    t0, n = time.time(), 0
    while time.time() - t0 < 5.0:
        net.feedforward(x)
        n += 512
    return dict(value=n / (time.time() - t0), unit="frames/s", cores=cores, kind="port", sample="numpy oracle, %d frames" % n)


def cpu_baseline_rbm(wl, tmp, frames=16384):
    """The reference's rbm-train-cd1-frmshuff (CPU build).  Its CPU path cannot sample Bernoulli hidden units
    (CuRand::BinarizeProbs never sizes its state there, cudamatrix/cu-rand-inl.h:162-168), so the timed model is the
    Gaussian-hidden twin of the workload -- same visible type, same sizes, same five products per bunch; the hidden
    sigmoid and the uniform draws are replaced by Gaussian noise.  Its own fps line (rbm-train-cd1-frmshuff.cc:174)."""
    exe = _ref_tool("rbm-train-cd1-frmshuff")
    cores = usable_cpus()
    if not os.path.exists(exe):
        return None
    dims = wl["dims"]
    twin = os.path.join(tmp, "cpu_rbm_twin.nnet")
    write_rbm(twin, dims[0], dims[1], wl["types"][0], "gauss", seed=1)
    fa, _, n = _synthetic_archive(tmp, dims, frames, 13, labels=False, unit_interval=wl["types"][0] == "bern")
    r = _run_ref([exe, "--bunchsize=%d" % BUNCH, "--cachesize=%d" % min(n, 32768), "--learn-rate=%g" % wl["lr"],
                  "--momentum=%g" % wl["mom"], "--l2-penalty=%g" % wl["l2"], "--binary=true", twin, "ark:" + fa,
                  os.path.join(tmp, "cpu_rbm_out.nnet")], cores, r"RBM TRAINING FINISHED (\S+)s, fps(\S+),")
    if not r:
        return None
    return dict(value=r[0], unit="frames/s", cores=cores, kind="reference",
                sample="%d frames, bunch %d, %s-gauss twin of the %s-%s model (the reference's CPU path cannot run Bernoulli "
                       "hidden units); the tool's own fps line (main loop %.1f s of %.1f s wall)" %
                       (n, BUNCH, wl["types"][0], wl["types"][0], wl["types"][1], r[1], r[2]),
                blas="OpenBLAS (scipy), %d threads" % cores)


class Workload:
    """One of WORKLOADS set up on the current GPU: the model file, the trainer and the resident synthetic data;
    step(i) is one pass of the hot path."""

    def __init__(self, name, h, nn, rank, world, tmp, loopback_env=True):
        self.name, self.h, self.nn = name, h, nn
        wl = self.wl = WORKLOADS[name]
        dims, kind = wl["dims"], wl["kind"]
        self.dims, self.kind = dims, kind
        self.model_path = os.path.join(tmp, "model_%s.nnet" % name)
        # same weights on every rank, like a shared model-in file
        if kind == "rbm":
            write_rbm(self.model_path, dims[0], dims[1], wl["types"][0], wl["types"][1], seed=1)
        else:
            write_model(self.model_path, dims, seed=1)
        rs = np.random.RandomState(100 + rank)
        self.batch_rows = None
        if kind == "forward":
            # utterance-sharded inference: every rank works through its own utterances, no collective.  As the tool
            # does (nnet-forward --batch-frames=4096): utterances gathered until a batch holds 4096 frames or more, one
            # pass through the network per batch, GEMM plans that keep a row's bits independent of its batch.
            self.trainer = nn.Trainer(self.model_path)
            h.call("hipK_gemm_row_invariant", 1)
            lens = np.random.RandomState(777 + rank).randint(300, 1201, 64)
            if os.environ.get("KALDI_BENCH_UTT_ROWS"):   # (probe: one "utterance" of this many rows per pass, free plans)
                lens = np.full(16, int(os.environ["KALDI_BENCH_UTT_ROWS"]))
                h.call("hipK_gemm_row_invariant", 0)
                batch_rows = [int(n) for n in lens]
            else:
                batch_rows, cur = [], 0
                for n in lens:
                    cur += int(n)
                    if cur >= 4096:
                        batch_rows.append(cur)
                        cur = 0
            self.batch_rows = batch_rows
            self.utts = [h.DevMatrix.from_numpy(rs.standard_normal((n, dims[0])).astype(np.float32)) for n in batch_rows]
            self.post = h.DevMatrix(max(batch_rows), dims[-1])
        else:
            if kind == "rbm":
                self.trainer = nn.RbmTrainer(self.model_path)
                self.trainer.set_train_opts(wl["lr"], wl["mom"], wl["l2"])
            else:
                self.trainer = nn.Trainer(self.model_path)
                self.trainer.set_train_opts(0.008)
                self.trainer.keep_posteriors(False)   # like the tool: only the loss and its gradient leave the output layer
                if world > 1:
                    self.trainer.set_data_parallel(world)
                elif loopback_env and os.environ.get("KALDI_BENCH_LOOPBACK_DP"):
                    # rehearsal of the data-parallel step on one GPU: gradient GEMM, (local) exchange, shared step
                    # (KALDI_BENCH_LOOPBACK_ONE_RANK: the stand-in does the device work of ONE of the n ranks -- its shard
                    # of the sums and of the step -- instead of playing all of them; timing only)
                    n = int(os.environ["KALDI_BENCH_LOOPBACK_DP"])
                    h.call("hipK_comm_init_loopback_one_rank" if os.environ.get("KALDI_BENCH_LOOPBACK_ONE_RANK") else "hipK_comm_init_loopback", n)
                    self.trainer.set_data_parallel(n)
            # a synthetic, already shuffled frame cache resident in HBM (each rank its own frames)
            cache_rows = 16 * BUNCH
            x = rs.standard_normal((cache_rows, dims[0])) if wl.get("types", ("gauss",))[0] == "gauss" else rs.random_sample((cache_rows, dims[0]))
            self.cache = h.DevMatrix.from_numpy(x.astype(np.float32))
            self.labels = h.DevMatrix.from_numpy(rs.randint(0, dims[-1], cache_rows).astype(np.int32), pitched=False)
            self.nb = cache_rows // BUNCH

    def frames_of(self, i):
        return self.batch_rows[i % len(self.utts)] if self.kind == "forward" else BUNCH

    def step(self, i):
        if self.kind == "forward":
            self.trainer.feedforward(self.utts[i % len(self.utts)], self.post)
            return
        b = i % self.nb
        rows = ctypes.c_void_p(self.cache.ptr.value + b * BUNCH * self.cache.stride * 4)
        if self.kind == "rbm":
            self.trainer.cd1_bunch(rows, self.cache.stride, BUNCH, self.dims[0])
        else:
            self.trainer.train_bunch(rows, self.cache.stride, BUNCH, self.dims[0],
                                     ctypes.c_void_p(self.labels.ptr.value + b * BUNCH * 4))

    def check(self, steps):
        """sanity figure of the run: mean cross-entropy of training (ln(classes) for random labels at the start), the
        posteriors' row sums of the last forward batch"""
        if self.kind == "forward":
            last = self.post.numpy()[:self.batch_rows[(steps - 1) % len(self.utts)]]
            return {"rows": int(last.shape[0]), "posterior_row_sum_min": round(float(last.sum(axis=1).min()), 5),
                    "posterior_row_sum_max": round(float(last.sum(axis=1).max()), 5)}
        frames, correct, loss = self.trainer.stats() if self.kind == "train" else (0, 0, 0.0)
        return {"frames": frames, "mean_xent": round(loss / max(frames, 1), 4)}

    def weights_digest(self):
        """(sum, sum of squares) over all weights and biases in float64: equal on replicas that hold the same bits"""
        tot, sq = 0.0, 0.0
        for i in range(len(self.dims) - 1):
            W, b = self.trainer.layer_params(2 * i, self.dims[i + 1], self.dims[i])
            tot += float(W.sum(dtype=np.float64)) + float(b.sum(dtype=np.float64))
            sq += float((W.astype(np.float64) ** 2).sum()) + float((b.astype(np.float64) ** 2).sum())
        return tot, sq

    def cpu_baseline(self, tmp, small=False):
        # bounded samples: ~15 s of CPU work for the workload the run is about, 2-6 s for each of the secondary ones
        big = len(self.dims) > 6 or self.dims[0] > 1024   # the Switchboard-sized net, the 2048 x 2048 RBM
        if self.kind == "train":
            return cpu_baseline(self.dims, self.model_path, tmp, (65536 if not big else 16384) if small else 32768, 0 if small else 4096)
        if self.kind == "forward":
            return cpu_baseline_forward(self.dims, self.model_path, tmp, (16384 if big else 65536) if small else (65536 if big else 262144))
        return cpu_baseline_rbm(self.wl, tmp, (16384 if big else 65536) if small else (65536 if big else 131072))

    def close(self):
        self.trainer.close()
        if self.kind == "forward":
            self.h.call("hipK_gemm_row_invariant", 0)


def measure(W, h, nn, fence, steps, warmup, setup_steps=40, sync=None):
    """set-up steps, a pass with every GEMM launch timed by HIP events bound to it (the roofline), warm-up, `steps` timed steps.
    -> dict of raw figures (this rank's)."""
    # Roofline of the dominant kernel, measured first: the same steps with a start and a stop event bound to every
    # GEMM launch on the stream it is queued on (the kernel's own begin and end stamps; a pass of its own: a timed
    # dispatch costs the host and the queue a little, which must not sit inside the timed region).  This pass also takes the GPU out of its idle clock state: from idle the shader clock needs
    # ~13-20 ms of load to climb from 2.14 to 2.40 GHz (tools/clock_ramp.hip), which is most of a 20-step run.
    roof_steps = max(30, min(steps, 50))
    for i in range(setup_steps):
        W.step(i)   # code objects, workspaces, first-touch, and ~28 ms of load: the clock is up when the events start
    fence()
    h.call("hipK_gemm_timing", 1)
    t1 = time.perf_counter()
    for i in range(roof_steps):
        W.step(i)
    cap = roof_steps * 64
    sp_start, sp_dur, sp_gf = (ctypes.c_double * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
    sp_n = ctypes.c_int()
    h.call("hipK_gemm_timing_spans", sp_start, sp_dur, sp_gf, cap, ctypes.byref(sp_n))
    ms, busy, launches, fl = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
    h.call("hipK_gemm_timing_read", ctypes.byref(ms), ctypes.byref(busy), ctypes.byref(launches), ctypes.byref(fl))
    roof_wall = time.perf_counter() - t1
    h.call("hipK_gemm_timing", 0)
    # reading the events back left the GPU idle for several milliseconds and its clock back in the idle state: the
    # same load as the set-up pass again, so that a short warm-up (the driver's 5 steps = 3 ms) starts at speed
    for i in range(setup_steps):
        W.step(i)
    fence()
    for i in range(warmup):
        W.step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        W.step(i)
    nn.synchronize()
    if sync:
        sync()
    elapsed = time.perf_counter() - t0
    return dict(elapsed=elapsed, roof_steps=roof_steps, roof_wall=roof_wall, gemm_ms=ms.value, gemm_busy_ms=busy.value,
                launches=launches.value, flops=fl.value, spans=(sp_n.value, sp_dur, sp_gf),
                frames=sum(W.frames_of(i) for i in range(steps)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="swbd", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="only the named workload (default with one GPU and the default workload: the other workloads of "
                         "the benchmark follow as short runs in a 'secondary' array)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" %
                     (args.gpus, args.gpus))
        args.gpus = world

    import torch  # plumbing only: rendezvous, barrier, max-over-ranks
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    rehearse_pg = world == 1 and os.environ.get("KALDI_BENCH_FORCE_PG") and "MASTER_ADDR" in os.environ
    if world > 1 or rehearse_pg:
        # (KALDI_BENCH_FORCE_PG: a one-rank group, to rehearse torch's RCCL streams beside the library's on one GPU)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist.barrier()

    from kaldi_amd import hip as h
    from kaldi_amd import nnet as nn
    h.init(local_rank)

    wl = WORKLOADS[args.workload]
    dims, kind = wl["dims"], wl["kind"]
    tmp = tempfile.mkdtemp(prefix="kaldi_amd_bench_")

    if world > 1 and kind == "train":
        # RCCL communicator of the device library: rank 0 creates the id, torch broadcasts the bytes
        uid = (ctypes.c_ubyte * h.COMM_ID_BYTES)()
        if rank == 0:
            h.call("hipK_comm_get_unique_id", uid)
        t_uid = torch.tensor(list(uid), dtype=torch.uint8, device="cuda")
        dist.broadcast(t_uid, 0)
        uid = (ctypes.c_ubyte * h.COMM_ID_BYTES)(*t_uid.cpu().tolist())
        h.call("hipK_comm_init", uid, rank, world)

    W = Workload(args.workload, h, nn, rank, world, tmp)

    def fence():
        nn.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    m = measure(W, h, nn, fence, args.steps, args.warmup, sync=torch.cuda.synchronize)
    elapsed = m["elapsed"]
    if world > 1:
        dist.barrier()
        t_el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        elapsed = float(t_el.item())
    roof_steps = m["roof_steps"]
    sp_n, sp_dur, sp_gf = m["spans"]

    check = W.check(args.steps)
    done = m["frames"]  # frames of this rank inside the timed region
    value = world * done / elapsed
    replicas = None
    if world > 1 and kind == "train":
        # synchronous data-parallel training keeps the replicas bit-identical: compare a digest of all parameters
        dg = torch.tensor(W.weights_digest(), dtype=torch.float64, device="cuda")
        lo, hi = dg.clone(), dg.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas = bool(torch.equal(lo, hi)) and bool(torch.isfinite(dg).all())

    # per kernel class: on one GPU the GEMM launches of a step come one after another in a fixed order --
    # forward layer by layer, top-down back-propagation, then the updates of all layers as one grouped launch
    # (with KALDI_AMD_NO_GROUP_UPDATES: back-propagation / update pairs, the bottom layer's update last)
    per_step = sp_n // roof_steps if roof_steps else 0
    classes = {}
    nl = len(dims) - 1
    FWD_OUT, FWD_HID = "forward (NT), output layer: split-K slabs for the softmax kernel", "forward (NT), hidden layer: bias + sigmoid epilogue"
    BP, UPD = "back-propagation (NN): y(1-y) epilogue", "weight update (TN): SGD step in the epilogue, bias in the launch"
    plain = world == 1 and not os.environ.get("KALDI_BENCH_LOOPBACK_DP")
    if kind == "train" and plain and per_step in (2 * nl, 3 * nl - 1) and sp_n == per_step * roof_steps:
        for k in range(sp_n):
            pos = k % per_step
            if pos < nl:
                name = FWD_OUT if pos == nl - 1 else FWD_HID
            elif per_step == 2 * nl:
                name = (UPD + ", all %d layers in one grouped launch" % nl if pos == per_step - 1
                        else BP + (", output layer" if pos == nl else ", hidden layer"))
            else:
                j = pos - nl
                upd = (j % 2 == 1) or pos == per_step - 1
                name = (UPD if upd else BP) + (", output layer" if j < 2 else ", hidden layer")
            c = classes.setdefault(name, [0, 0.0, 0.0])
            c[0] += 1
            c[1] += sp_dur[k] * 1e3
            c[2] += sp_gf[k]

    if rank == 0:
        fpf = flops_per_frame(dims, kind)
        ms_v, busy_v, launches_v, fl_v = m["gemm_ms"], m["gemm_busy_ms"], m["launches"], m["flops"]
        # achieved = algorithmic FLOPs / time the GEMM kernels occupy the chip (overlapping launches
        # counted once); per_launch = the same FLOPs / summed launch durations
        achieved = fl_v / (busy_v * 1e-3) / 1e12 if busy_v > 0 else 0.0
        per_launch = fl_v / (ms_v * 1e-3) / 1e12 if ms_v > 0 else 0.0
        # HBM-side bytes per GEMM launch come from rocprofv3 PMC passes of this same command (their
        # summary is committed; a PMC pass cannot run inside the timed process)
        traffic, traffic_source = None, None
        tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "gemm_traffic.json")
        if args.workload == "swbd" and os.path.exists(tf):
            traffic = json.load(open(tf)).get("gemm_hbm_bytes_per_launch")
            traffic_source = "profiles/gemm_traffic.json (a committed rocprofv3 PMC pass of this command, not a measurement of this run)"
        out = {
            "metric": "%s, %s" % (METRICS[kind], args.workload),
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "bunch_per_gpu": BUNCH if kind != "forward" else None,
                       "global_bunch": BUNCH * world if kind != "forward" else None,
                       "frames_per_step": round(done / args.steps, 1),
                       "parallelism": ("single" if world == 1 else "dp%d" % world if kind == "train" else "replicas%d" % world),
                       # (Nnet::SetDataParallel's rule: rows gathered for 2-4 ranks, sharded gradient sums above)
                       # one-GPU rehearsal of the data-parallel step: what the stand-in communicator did
                       "rehearsal": ({"stand_in_ranks": int(os.environ["KALDI_BENCH_LOOPBACK_DP"]),
                                      "exchange": os.environ.get("KALDI_AMD_DP_EXCHANGE") or
                                      ("gather" if int(os.environ["KALDI_BENCH_LOOPBACK_DP"]) <= 4 else "shard"),
                                      "stand_in": ("one rank's device work (timing only: the other shards are never stepped)"
                                                   if os.environ.get("KALDI_BENCH_LOOPBACK_ONE_RANK") else
                                                   "every rank's sums and steps on this GPU (the whole job's arithmetic)")}
                                     if world == 1 and kind == "train" and os.environ.get("KALDI_BENCH_LOOPBACK_DP") else None),
                       "exchange": (None if world == 1 or kind != "train" else
                                    os.environ.get("KALDI_AMD_DP_EXCHANGE") or ("gather" if world <= 4 else "shard")),
                       "flops_per_frame": fpf,
                       "pct_fp32_mfma_peak_end_to_end": round(100.0 * value * fpf / world / (F32_MFMA_PEAK_TFLOPS * 1e12), 2)},
            # frac = algorithmic FLOPs of the GEMM launches / summed launch durations / peak: what the per-kernel
            # averages of a rocprofv3 kernel trace of this command reproduce (profiles/); on one GPU the launches
            # of a step run one after another, so this is also FLOPs over the time a GEMM kernel is running
            "roofline": {"bound": "mfma", "kernel": "gemm_kernel (fp32 v_mfma_f32_32x32x2_f32, all instantiations)",
                         "achieved": round(per_launch, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(per_launch / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "launches_per_step": launches_v / roof_steps,
                         "avg_launch_us": round(ms_v / max(launches_v, 1) * 1e3, 2),
                         "gflop_per_launch": round(fl_v / max(launches_v, 1) * 1e-9, 3),
                         "gemm_launch_ms_sum_per_step": round(ms_v / roof_steps, 4),
                         "gemm_busy_ms_per_step": round(busy_v / roof_steps, 4),
                         "achieved_over_busy_time": round(achieved, 2),
                         "flops_per_step": fl_v / roof_steps,
                         "instrumented_pass_ms_per_step": round(m["roof_wall"] / roof_steps * 1e3, 4),
                         "per_kernel": [{"kernel": k, "launches_per_step": round(v[0] / roof_steps, 2),
                                         "avg_launch_us": round(v[1] / v[0], 2), "gflop_per_launch": round(v[2] / v[0], 3),
                                         "achieved": round(v[2] / v[1] * 1e3 if v[1] else 0.0, 2),
                                         "frac": round(v[2] / v[1] * 1e3 / F32_MFMA_PEAK_TFLOPS if v[1] else 0.0, 4)}
                                        for k, v in sorted(classes.items())]},
            "order": "40 set-up steps (first touch; the clock climbs out of its idle state), %d steps with HIP events bound to every GEMM launch (roofline), 40 steps to bring the clock back up after the events were read, %d warm-up steps, %d timed steps" % (roof_steps, args.warmup, args.steps),
            "check": check,
        }
        if replicas is not None:
            out["check"]["replicas_identical"] = replicas
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = W.cpu_baseline(tmp)

    W.close()
    if world > 1 and kind == "train":
        h.call("hipK_comm_destroy")

    # The other workloads of the benchmark (RM-sized training, both forward passes, both RBMs), as short runs of the same
    # kind in the same process: one record per configuration from one invocation.  The headline above is measured first
    # and is not affected.
    if (rank == 0 and world == 1 and args.workload == "swbd" and not args.no_secondary and
            not os.environ.get("KALDI_BENCH_LOOPBACK_DP")):
        secondary = []
        for name in ("rm", "swbd-forward", "rm-forward", "rbm-gauss-bern", "rbm-bern-bern"):
            W2 = Workload(name, h, nn, 0, 1, tmp, loopback_env=False)
            k2 = W2.kind
            steps2 = 30 if name == "swbd-forward" else 100 if k2 == "forward" else 300
            m2 = measure(W2, h, nn, lambda: nn.synchronize(), steps2, 10, setup_steps=20)
            fpf2 = flops_per_frame(W2.dims, k2)
            v2 = m2["frames"] / m2["elapsed"]
            e = {"workload": W2.wl["name"], "metric": "%s, %s" % (METRICS[k2], name), "value": round(v2, 1), "unit": "frames/s",
                 "steps": steps2, "ms_per_step": round(m2["elapsed"] / steps2 * 1e3, 4),
                 "frames_per_step": round(m2["frames"] / steps2, 1), "flops_per_frame": fpf2,
                 # end to end: algorithmic FLOPs of the step over its wall time; gemm_launches: over the summed GEMM launch durations
                 "frac": round(v2 * fpf2 / (F32_MFMA_PEAK_TFLOPS * 1e12), 4),
                 "frac_gemm_launches": round(m2["flops"] / (m2["gemm_ms"] * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4) if m2["gemm_ms"] > 0 else None,
                 "check": W2.check(steps2)}
            if not args.no_cpu_baseline:
                e["cpu_baseline"] = W2.cpu_baseline(tmp, small=True)
            W2.close()
            secondary.append(e)
        out["secondary"] = secondary

    if rank == 0:
        print(json.dumps(out), flush=True)

    if rehearse_pg:
        t_el = torch.ones(4, device="cuda")
        dist.all_reduce(t_el)
    if world > 1 or rehearse_pg:
        dist.barrier()
        dist.destroy_process_group()
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
