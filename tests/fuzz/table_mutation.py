"""Damaged tables never become memory errors: tests/fuzz/table_mutation.cc (the readers the tools use, nothing else) is
built with the address and undefined-behaviour sanitizers from the host sources and run on archives and script files
with flipped, cut, inserted and overwritten bytes.  Allowed ends: status 0 (read to the end, whatever the values),
1 (a reported error), an abort from an assertion of the library (counted on its own); a sanitizer report, a
signal or a run that does not end is a finding.  Needs no GPU.

usage: python tests/fuzz/table_mutation.py [--cases N] [--seed S] [--out file]"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import nnet1_oracle as O  # noqa: E402

SRC = os.path.join(ROOT, "kaldi_amd", "csrc")
EXE = os.path.join(ROOT, "build", "asan", "table_mutation")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    srcs = [os.path.join(ROOT, "tests", "fuzz", "table_mutation.cc")] + [
        os.path.join(SRC, f) for f in ("base/kaldi-error.cc", "base/kaldi-math.cc", "base/io-funcs.cc",
                                       "util/kaldi-io.cc", "util/kaldi-table.cc")]
    deps = srcs + [os.path.join(SRC, d, f) for d in ("base", "util", "matrix") for f in os.listdir(os.path.join(SRC, d))]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) > os.path.getmtime(s) for s in deps):
        return
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-pthread", "-Wno-deprecated-declarations", "-I" + SRC, "-I" + os.path.join(ROOT, "include"),
                    "-o", EXE] + srcs, check=True)


def text_matrix_ark(utts):
    out = []
    for k, m in utts:
        out.append("%s  [\n" % k + "\n".join("  " + " ".join("%g" % v for v in row) for row in m) + " ]\n")
    return "".join(out).encode()


def text_ali_ark(alis):
    return "".join("%s %s\n" % (k, " ".join(str(int(v)) for v in a)) for k, a in alis).encode()


def binary_ali_ark(alis):
    out = b""
    for k, a in alis:
        out += k.encode() + b" \0B" + b"\x04" + np.int32(len(a)).tobytes() + b"".join(b"\x04" + np.int32(v).tobytes() for v in a)
    return out


def mutate(rs, data):
    b = bytearray(data)
    for _ in range(int(rs.randint(1, 4))):
        how = rs.randint(0, 6)
        if not b:
            break
        p = int(rs.randint(0, len(b)))
        if how == 0:
            b[p] ^= 1 << int(rs.randint(0, 8))
        elif how == 1:
            del b[p:]
        elif how == 2:
            b[p:p] = bytes(rs.randint(0, 256, int(rs.randint(1, 9))).astype(np.uint8))
        elif how == 3:
            del b[p:p + int(rs.randint(1, 17))]
        elif how == 4:  # a count or dimension field made huge or negative
            b[p:p + 4] = np.int32(rs.choice([-1, -2 ** 31, 2 ** 31 - 1, 2 ** 24, 0])).tobytes()
        else:
            b[p] = int(rs.choice([0, 10, 32, 91, 93, 255]))
    return bytes(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=600)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    build()
    rs = np.random.RandomState(args.seed)
    utts = [("utt%d" % i, rs.randn(int(rs.randint(1, 9)), 5).astype(np.float32)) for i in range(6)]
    alis = [("utt%d" % i, rs.randint(0, 50, int(rs.randint(1, 12)))) for i in range(6)]
    vecs = b"".join(k.encode() + b" \0BFV \x04" + np.int32(7).tobytes() + rs.randn(7).astype(np.float32).tobytes() for k, _ in utts)
    seeds = [("feats", "ark", O.write_matrix_ark(utts)), ("feats", "ark", text_matrix_ark(utts)),
             ("ali", "ark", text_ali_ark(alis)), ("ali", "ark", binary_ali_ark(alis)),
             ("ali_random", "ark", text_ali_ark(alis)), ("ali_random", "ark", binary_ali_ark(alis)),
             ("ali_random", "ark,s", text_ali_ark(alis)), ("vec", "ark", vecs)]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:exitcode=98:print_stacktrace=1")
    counts, findings = {}, []
    with tempfile.TemporaryDirectory() as tmp:
        # the undamaged tables must read to the end
        for i, (mode, form, data) in enumerate(seeds):
            f = os.path.join(tmp, "good%d" % i)
            open(f, "wb").write(data)
            r = subprocess.run([EXE, mode, "%s:%s" % (form, f)], capture_output=True, timeout=60, env=env)
            if r.returncode != 0:
                findings.append("undamaged %s table %d: status %d\n%s" % (mode, i, r.returncode, r.stderr.decode(errors="replace")[-400:]))
        for case in range(args.cases):
            mode, form, data = seeds[int(rs.randint(0, len(seeds)))]
            f = os.path.join(tmp, "t.ark")
            open(f, "wb").write(mutate(rs, data))
            spec = "%s:%s" % (form, f)
            if rs.rand() < 0.2:  # through a script file with offsets, some of them wrong
                lines = ["utt%d %s:%d\n" % (i, f, int(rs.randint(0, len(data) + 8))) for i in range(4)]
                open(os.path.join(tmp, "t.scp"), "w").write("".join(lines))
                spec = "scp:" + os.path.join(tmp, "t.scp")
            try:
                r = subprocess.run([EXE, mode, spec], capture_output=True, timeout=60, env=env)
            except subprocess.TimeoutExpired:
                findings.append("case %d (%s %s): does not end" % (case, mode, spec))
                open(os.path.join(ROOT, "gpurun_out", "table_mutation_case%d.bin" % case), "wb").write(open(f, "rb").read())
                continue
            err = r.stderr.decode(errors="replace")
            if "Sanitizer" in err or "runtime error:" in err or r.returncode in (98, 99):
                kind = "sanitizer"
            elif r.returncode == 0:
                kind = "read to the end"
            elif r.returncode == 1:
                kind = "reported error"
            elif r.returncode == -6:
                kind = "library assertion"
            else:
                kind = "status %d" % r.returncode
            counts[kind] = counts.get(kind, 0) + 1
            if kind not in ("read to the end", "reported error", "library assertion"):
                findings.append("case %d (%s %s): %s\n%s" % (case, mode, spec, kind, err[-1500:]))
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                open(os.path.join(ROOT, "gpurun_out", "table_mutation_case%d.bin" % case), "wb").write(open(f, "rb").read())
    lines = ["table_mutation: %d damaged tables, %d findings" % (args.cases, len(findings))]
    lines += ["  %-20s %d" % kv for kv in sorted(counts.items())]
    lines += findings
    text = "\n".join(lines)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
