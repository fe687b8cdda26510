"""Multi-rank logic on the CPU (gloo, world size 2): what the N>1 paths rely on, checked without a GPU.

  * data-parallel training: every rank computes the raw gradient sums of its own bunch, the sums are
    all-reduced, every rank applies the same step with the total frame count.  With the oracle's
    arithmetic on both sides this must equal one process training on the concatenated bunch (the
    property the RCCL path of Nnet::SetDataParallel is built on; the GPU form of the same check is
    tests/test_gpu_dist.py).
  * sharded nnet-forward: every rank evaluates the utterance -> rank rule on its own while streaming
    the same table; the ranks must agree, cover every utterance exactly once and balance frames.
  * per-cache bunch-count agreement (min over ranks), as used by the data-parallel training tool.
  * the sharded exchange (KALDI_AMD_DP_EXCHANGE=shard): the gradient laid out like the pitched weight memory,
    reduce-scatter, the step on the rank's own shard only, all-gather of the new weights -- must leave every
    rank with the bits of the all-reduce form.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nnet1_oracle as O  # noqa: E402

f32 = np.float32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_net(rs, dims):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(O.BiasedLinearity((0.3 * rs.standard_normal((dims[i + 1], dims[i]))).astype(f32),
                                        (0.1 * rs.standard_normal(dims[i + 1])).astype(f32)))
        layers.append(O.Sigmoid(dims[i + 1]) if i + 2 < len(dims) else O.Softmax(dims[i + 1]))
    return O.Nnet(layers)


def _dp_step(net, x, tgt, world, allreduce, lr, mom, avg):
    """One synchronous data-parallel step in the oracle's arithmetic (gradient sums -> all-reduce ->
    the update of nnet-biasedlinearity.h:71-102 with the global frame count)."""
    post = net.propagate(x)
    err, _ = O.diff_xent(tgt, post)
    buf = net.propagate_buf
    L = len(net.layers)
    cur = err
    for i in range(L - 1, -1, -1):
        l = net.layers[i]
        below = l.backpropagate(cur, buf[i], buf[i + 1]) if i > 0 else None
        if l.updatable:
            gW = O.gemm(1.0, cur, True, buf[i], False, 0.0, None)
            gb = O.add_row_sum_mat(1.0, cur, 0.0, np.zeros(l.dim_out, f32))
            flat = np.concatenate([gW.ravel(), gb]).astype(f32)
            flat = allreduce(flat)
            gW, gb = flat[:gW.size].reshape(gW.shape), flat[gW.size:]
            n_total = x.shape[0] * world
            g = f32(1.0 / n_total) if avg else f32(1.0)
            l.W_corr = (g * gW + f32(mom) * l.W_corr).astype(f32)
            l.b_corr = (g * gb + f32(mom) * l.b_corr).astype(f32)
            l.W = (f32(-lr) * l.W_corr + l.W).astype(f32)
            l.b = (f32(-lr) * l.b_corr + l.b).astype(f32)
        cur = below
    return post


def _dp_step_sharded(net, x, tgt, world, rank, reduce_to, allgather, allreduce, lr, mom, avg, state):
    """The same step with the sharded exchange of Nnet::Backpropagate: per layer the weight memory (rows padded
    to a multiple of 64 floats, as hipK_malloc_pitch lays them out) is cut into `world` shards; shard r of the
    gradient is summed onto rank r, which steps only those elements; the new weights are then gathered."""
    post = net.propagate(x)
    err, _ = O.diff_xent(tgt, post)
    buf = net.propagate_buf
    cur = err
    for i in range(len(net.layers) - 1, -1, -1):
        l = net.layers[i]
        below = l.backpropagate(cur, buf[i], buf[i + 1]) if i > 0 else None
        if l.updatable:
            rows, cols = l.W.shape
            stride = (cols + 63) // 64 * 64
            span = rows * stride
            assert span % world == 0 and (span // world) % 4 == 0
            per = span // world
            if i not in state:   # pitched images of the weights and of the momentum state
                wm, cm = np.zeros((rows, stride), f32), np.zeros((rows, stride), f32)
                wm[:, :cols] = l.W
                state[i] = (wm.reshape(-1), cm.reshape(-1))
            wmem, cmem = state[i]
            g2 = np.zeros((rows, stride), f32)
            g2[:, :cols] = O.gemm(1.0, cur, True, buf[i], False, 0.0, None)
            gflat = g2.reshape(-1)
            gb = allreduce(O.add_row_sum_mat(1.0, cur, 0.0, np.zeros(l.dim_out, f32)))
            mine = None
            for r in range(world):   # reduce-scatter: shard r is summed onto rank r
                got = reduce_to(gflat[r * per:(r + 1) * per], r)
                if r == rank:
                    mine = got
            g = f32(1.0 / (x.shape[0] * world)) if avg else f32(1.0)
            sl = slice(rank * per, (rank + 1) * per)
            cmem[sl] = (g * mine + f32(mom) * cmem[sl]).astype(f32)
            new_shard = (f32(-lr) * cmem[sl] + wmem[sl]).astype(f32)
            wmem[:] = allgather(new_shard)             # every rank now holds the whole new weight memory
            l.W = wmem.reshape(rows, stride)[:, :cols].copy()
            l.b_corr = (g * gb + f32(mom) * l.b_corr).astype(f32)
            l.b = (f32(-lr) * l.b_corr + l.b).astype(f32)
        cur = below
    return post


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        dims = [20, 24, 16, 9]
        lr, mom, avg, nsteps, nb = 0.05, 0.5, True, 3, 32
        rs = np.random.RandomState(7)
        net = _make_net(rs, dims)
        X = rs.standard_normal((nsteps, world * nb, dims[0])).astype(f32)
        T = rs.randint(0, dims[-1], (nsteps, world * nb)).astype(np.int32)

        def allreduce(v):
            t = torch.from_numpy(v.copy())
            dist.all_reduce(t)
            return t.numpy()

        for s in range(nsteps):
            xs, ts = X[s, rank * nb:(rank + 1) * nb], T[s, rank * nb:(rank + 1) * nb]
            _dp_step(net, xs, ts, world, allreduce, lr, mom, avg)
        params = np.concatenate([np.concatenate([l.W.ravel(), l.b]) for l in net.layers if l.updatable])

        # the same three steps with the sharded exchange
        def reduce_to(v, dst):
            t = torch.from_numpy(v.copy())
            dist.reduce(t, dst)
            return t.numpy()

        def allgather(v):
            parts = [torch.zeros(v.size, dtype=torch.float32) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(v.copy()))
            return np.concatenate([p_.numpy() for p_ in parts])

        net2, state = _make_net(np.random.RandomState(7), dims), {}
        for s in range(nsteps):
            xs, ts = X[s, rank * nb:(rank + 1) * nb], T[s, rank * nb:(rank + 1) * nb]
            _dp_step_sharded(net2, xs, ts, world, rank, reduce_to, allgather, allreduce, lr, mom, avg, state)
        params_sharded = np.concatenate([np.concatenate([l.W.ravel(), l.b]) for l in net2.layers if l.updatable])

        # sharded forward: every rank evaluates the rule itself from the same table
        import kaldi_amd.nnet as nn
        lens = np.random.RandomState(3).randint(1, 900, 257)
        mine = nn.shard_assign(world, lens)
        gathered = [torch.zeros(lens.size, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(mine.copy()))
        agree = all(bool((g.numpy() == mine).all()) for g in gathered)
        my_frames = int(lens[mine == rank].sum())

        # bunch-count agreement: every rank stops after the smallest count
        counts = torch.tensor([17 + 5 * rank], dtype=torch.int64)
        dist.all_reduce(counts, op=dist.ReduceOp.MIN)
        q.put((rank, params, agree, my_frames, int(lens.sum()), int(counts.item()), params_sharded))
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_equals_single_process_and_shards_agree():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    # single process on the concatenated bunches, the oracle's own update
    dims = [20, 24, 16, 9]
    lr, mom, nsteps, nb = 0.05, 0.5, 3, 32
    rs = np.random.RandomState(7)
    net = _make_net(rs, dims)
    net.set_train_opts(lr, mom, average_grad=True)
    X = rs.standard_normal((nsteps, world * nb, dims[0])).astype(f32)
    T = rs.randint(0, dims[-1], (nsteps, world * nb)).astype(np.int32)
    for s in range(nsteps):
        post = net.propagate(X[s])
        net.backpropagate(O.diff_xent(T[s], post)[0])
    ref = np.concatenate([np.concatenate([l.W.ravel(), l.b]) for l in net.layers if l.updatable])

    np.testing.assert_array_equal(res[0][1], res[1][1])                      # replicas stay identical
    np.testing.assert_array_equal(res[0][6], res[1][6])                      # also with the sharded exchange,
    np.testing.assert_array_equal(res[0][6], res[0][1])                      # which gives the all-reduce form's bits
    np.testing.assert_allclose(res[0][1], ref, rtol=2e-5, atol=2e-6)         # and match one big bunch
    assert res[0][2] and res[1][2]                                            # same shard map on every rank
    total = res[0][4]
    assert res[0][3] + res[1][3] == total                                     # every frame exactly once
    assert abs(res[0][3] - res[1][3]) <= 900                                  # balanced to one utterance
    assert res[0][5] == res[1][5] == 17                                       # min over ranks


def test_shard_rule_and_job_substitution():
    import kaldi_amd.nnet as nn
    lens = [10, 10, 5, 1, 30, 2]
    assert nn.shard_assign(1, lens).tolist() == [0] * 6
    # fewest frames so far, lowest rank on ties
    assert nn.shard_assign(2, lens).tolist() == [0, 1, 0, 1, 1, 0]
    assert nn.shard_assign(3, []).tolist() == []
    assert nn.substitute_job("ark:out.JOB.ark", 2) == "ark:out.3.ark"
    assert nn.substitute_job("ark,scp:a.JOB.ark,a.JOB.scp", 0) == "ark,scp:a.1.ark,a.1.scp"
    assert nn.substitute_job("ark:plain.ark", 5) == "ark:plain.ark"


def test_comm_nonce_and_id_file(tmp_path, monkeypatch):
    """The RCCL id file of the tools (nnetbin/dist-utils.h): with a run id the nonce is the same whatever the
    parent process (ranks on other nodes, ranks started from several shells); without one the parent pid
    separates launches; a reader accepts only a complete file that carries its own launch's nonce."""
    import kaldi_amd.nnet as nn
    for var in ("KALDI_AMD_RUN_ID", "TORCHELASTIC_RUN_ID", "MASTER_PORT"):
        monkeypatch.delenv(var, raising=False)
    # no run id: the parent process is the launch
    assert nn.comm_nonce(100) != nn.comm_nonce(200)
    assert nn.comm_nonce(100) == "-:-:100"
    monkeypatch.setenv("MASTER_PORT", "29512")
    assert nn.comm_nonce(100) == "-:29512:100"
    # a launcher's run id: the same nonce under different parents
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "elastic7")
    assert nn.comm_nonce(100) == nn.comm_nonce(200) == "elastic7:29512"
    # KALDI_AMD_RUN_ID wins over the launcher's
    monkeypatch.setenv("KALDI_AMD_RUN_ID", "byhand")
    assert nn.comm_nonce(1) == nn.comm_nonce(99999) == "byhand:29512"

    path = str(tmp_path / "comm_id")
    ident = bytes(range(128))
    assert nn.comm_id_file_read(path, "byhand:29512", 128) is None            # absent: not yet
    nn.comm_id_file_write(path, "byhand:29512", ident)
    assert nn.comm_id_file_read(path, "byhand:29512", 128) == ident
    assert nn.comm_id_file_read(path, "other:29512", 128) is None             # another launch's file
    assert nn.comm_id_file_read(path, "byhand:2951", 128) is None
    assert (os.stat(path).st_mode & 0o777) == 0o600
    # a stale file is replaced, a truncated one is "not yet"
    nn.comm_id_file_write(path, "next:1", ident[::-1])
    assert nn.comm_id_file_read(path, "byhand:29512", 128) is None
    assert nn.comm_id_file_read(path, "next:1", 128) == ident[::-1]
    whole = open(path, "rb").read()
    open(path, "wb").write(whole[:-5])
    assert nn.comm_id_file_read(path, "next:1", 128) is None


def test_sharded_table_read_detects_truncation_inside_a_skipped_entry(tmp_path):
    """Every rank of the sharded nnet-forward walks the same archive and skips the payload of other ranks'
    utterances unread.  A table cut off inside an entry must be an error on EVERY rank -- also on the ranks that
    skip that entry -- on a file (seek) and on a pipe (discard), like an unfiltered read of the same table."""
    import kaldi_amd.nnet as nn
    rs = np.random.RandomState(5)
    lens = [30, 20, 10, 25, 15, 40]
    utts = [("utt%d" % i, rs.standard_normal((n, 7)).astype(f32)) for i, n in enumerate(lens)]
    whole = O.write_matrix_ark(utts)
    ark = str(tmp_path / "feats.ark")
    open(ark, "wb").write(whole)
    owner = nn.shard_assign(2, lens).tolist()
    per_utt = [n * 7 * 4 for n in lens]
    for spec in ("ark:" + ark, "ark:cat %s |" % ark):
        tot_kept = 0
        for rank in (0, 1):
            kept, frames, payload = nn.table_scan_shard(spec, 2, rank)
            mine = [i for i, r in enumerate(owner) if r == rank]
            assert kept == len(mine) and frames == sum(lens[i] for i in mine)
            if spec.endswith("|"):
                assert payload == sum(per_utt)                       # a pipe cannot be skipped, only discarded
            else:
                assert payload == sum(per_utt[i] for i in mine)      # a file: only the own payload is read
            tot_kept += kept
        assert tot_kept == len(lens)
    # cut the archive inside the last entry's payload: the rank that skips it must fail like the one that owns it
    cut = str(tmp_path / "cut.ark")
    open(cut, "wb").write(whole[:-40])
    for spec in ("ark:" + cut, "ark:cat %s |" % cut):
        for rank in (0, 1):
            with pytest.raises(nn.NnetError):
                nn.table_scan_shard(spec, 2, rank)
        with pytest.raises(nn.NnetError):
            nn.table_scan_shard(spec, 1, 0)
