#!/usr/bin/env python3
"""rbm-train-cd1-frmshuff of this build beside the reference's (oracle/_ref/bin, CPU) on the unit types the reference's
CPU branch can run (Gaussian hidden units: its Bernoulli sampling never sizes the state matrix there).  The two draw
their noise from different generators, so weights cannot agree element by element; what must agree is what does not
depend on the draw: the cache lines, the frame counts, the reconstruction error of the run to a few percent, and the
size of the weight change (Frobenius norm of W_out - W_in) to a few percent.

    python tests/fuzz/rbm_tool_stat_check.py [--out FILE]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nnet1_oracle as O  # noqa: E402

f32 = np.float32
REF = os.path.join(ROOT, "oracle", "_ref", "bin")
OURS = os.path.join(ROOT, "kaldi_amd", "bin")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rs = np.random.RandomState(21)
    tmp = tempfile.mkdtemp(prefix="rbmstat_")
    lines, bad = [], 0
    for case, (vis_type, vis, hid, opts) in enumerate([
            ("gauss", 40, 64, ["--learn-rate=0.001", "--momentum=0.5", "--l2-penalty=0.0002"]),
            ("bern", 64, 32, ["--learn-rate=0.01", "--momentum=0.0"]),
            ("gauss", 24, 100, ["--learn-rate=0.002", "--momentum=0.9", "--bunchsize=32", "--cachesize=4096"]),
            ("bern", 33, 17, ["--learn-rate=0.02", "--momentum=0.3", "--l2-penalty=0.001", "--bunchsize=100", "--cachesize=1000"])]):
        W0 = (0.05 * rs.standard_normal((hid, vis))).astype(f32)
        rbm = O.Rbm(W0, np.zeros(vis, f32), np.zeros(hid, f32), vis_type, "gauss")
        open(tmp + "/r.rbm", "wb").write(O.write_nnet([rbm], True))
        # data with structure: a few latent directions plus noise (probabilities for Bernoulli visible units)
        n_utts, basis = 40, rs.standard_normal((5, vis))
        utts = []
        for i in range(n_utts):
            z = rs.standard_normal((200 + 10 * i, 5)) @ basis + 0.3 * rs.standard_normal((200 + 10 * i, vis))
            utts.append(("utt%03d" % i, (1.0 / (1.0 + np.exp(-z)) if vis_type == "bern" else z).astype(f32)))
        open(tmp + "/f.ark", "wb").write(O.write_matrix_ark(utts))
        got = {}
        for who, d in (("ref", REF), ("ours", OURS)):
            out = "%s/o_%s.rbm" % (tmp, who)
            r = subprocess.run([os.path.join(d, "rbm-train-cd1-frmshuff")] + opts + [tmp + "/r.rbm", "ark:" + tmp + "/f.ark", out],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            err = r.stderr.decode(errors="replace")
            W = O.read_nnet(open(out, "rb").read())[0] if r.returncode == 0 else None
            caches = [re.sub(r"\s+", " ", l).strip() for l in err.splitlines() if l.startswith("Cache #")]
            # (the tool's own Mse line reports "frames:0": it prints a progress object that never flushes its counters;
            # the reconstruction error is computed here instead, mean field, on the first utterances)
            mse = None
            if W is not None:
                x = np.concatenate([m for _, m in utts[:8]]).astype(np.float64)
                rec = W.reconstruct(W.propagate(x.astype(f32)))
                mse = (None, str(x.shape[0]), "%.6g" % float(np.mean(np.sum((rec - x) ** 2, axis=1))))
                W = W.W
            got[who] = (r.returncode, W, caches, mse, err)
        (rc_r, W_r, c_r, m_r, e_r), (rc_o, W_o, c_o, m_o, e_o) = got["ref"], got["ours"]
        why = []
        if rc_r != 0 or rc_o != 0:
            why.append("exit %s / %s" % (rc_r, rc_o))
        else:
            if c_r != c_o:
                why.append("cache lines differ: %s / %s" % (c_r[-1:], c_o[-1:]))
            if (m_r is None) != (m_o is None) or (m_r and m_r[1] != m_o[1]):
                why.append("frames %s / %s" % (m_r, m_o))
            elif m_r and abs(float(m_r[2]) - float(m_o[2])) > 0.03 * abs(float(m_r[2])):
                why.append("err/frm %s / %s" % (m_r[2], m_o[2]))
            d_r, d_o = np.linalg.norm(W_r - W0), np.linalg.norm(W_o - W0)
            if abs(d_r - d_o) > 0.05 * d_r:
                why.append("|W - W0| %.4g / %.4g" % (d_r, d_o))
            # the two end points are two draws around the same expected update: closer to each other than to the start
            if np.linalg.norm(W_r - W_o) > 0.8 * d_r:
                why.append("|W_ref - W_ours| %.4g against |W_ref - W0| %.4g" % (np.linalg.norm(W_r - W_o), d_r))
        bad += 1 if why else 0
        line = "case %d %s-gauss %dx%d %s: " % (case, vis_type, vis, hid, " ".join(opts))
        if rc_r == 0 and rc_o == 0:
            line += "err/frm %s / %s, frames %s, |W-W0| %.4g / %.4g, |W_ref-W_ours| %.4g, %d cache lines  %s" % (
                m_r[2] if m_r else None, m_o[2] if m_o else None, m_r[1] if m_r else None, np.linalg.norm(W_r - W0), np.linalg.norm(W_o - W0),
                np.linalg.norm(W_r - W_o), len(c_r), ("<<< " + "; ".join(why)) if why else "ok")
        else:
            line += "<<< " + "; ".join(why) + "\n" + e_r[-400:] + "\n" + e_o[-400:]
        lines.append(line)
        print(line, flush=True)
    lines.insert(0, "rbm_tool_stat_check: %d cases, %d differ" % (len(lines), bad))
    print(lines[0])
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
