import os, sys, subprocess, numpy as np
sys.path.insert(0, '.')
from oracle import nnet1_oracle as O
keep = sys.argv[1]
for case in sorted(os.listdir(keep)):
    d = os.path.join(keep, case)
    cmd = open(os.path.join(d, "cmd.txt")).read().split("\n")
    tool, args = cmd[0], [a.replace(os.path.dirname(a.split(":",1)[-1]) if False else a, a) for a in cmd[1:]]
    # the kept directory was copied: rewrite the temporary directory in the paths
    old = None
    for a in args:
        if "diff_fuzz_" in a:
            i = a.index("/tmp/diff_fuzz_"); j = a.index("/", i + 5)
            old = a[i:j]; break
    args = [a.replace(old, d) for a in args]
    W = {}
    for name, exe, extra in (("ref", "oracle/_ref/bin/" + tool, []), ("fused", "kaldi_amd/bin/" + tool, []), ("unfused", "kaldi_amd/bin/" + tool, ["--fused=false"])):
        out = os.path.join(d, "look_%s.nnet" % name)
        r = subprocess.run([exe] + extra + args + [out], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        W[name] = [l for l in O.read_nnet(open(out, "rb").read()) if isinstance(l, O.BiasedLinearity)]
    print(case, tool, [a for a in args if a.startswith("--")])
    for k in range(len(W["ref"])):
        a, b, c = W["ref"][k].W, W["fused"][k].W, W["unfused"][k].W
        print("  layer %d: max|W| %.4g  |ref-fused| %.3g  |ref-unfused| %.3g  |fused-unfused| %.3g  elements off by >1e-4: %d / %d / %d of %d" % (
            k, np.abs(a).max(), np.abs(a - b).max(), np.abs(a - c).max(), np.abs(b - c).max(),
            (np.abs(a - b) > 1e-4).sum(), (np.abs(a - c) > 1e-4).sum(), (np.abs(b - c) > 1e-4).sum(), a.size))
