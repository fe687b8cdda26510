"""Extracts what the recipes' scheduler passes to, and reads from, the trainer -- the option names of its
cross-validation and training command lines and the awk program that takes the accuracy from the log -- from
egs/aurora4/a00/steps/nnet_scheduler/train_scheduler.sh:76-120 of the reference, into scheduler_contract.json.
Strings only: the tests that use the fixture (and the one that re-derives it where the reference tree exists)
never need the script itself.

    python tests/golden/make_scheduler_contract.py [/root/reference]
"""
import json
import os
import re
import sys

REL = "egs/aurora4/a00/steps/nnet_scheduler/train_scheduler.sh"


def extract(ref_root):
    text = open(os.path.join(ref_root, REL)).read()
    # the three invocations of $train_tool: the pre-run, the training run and the per-iteration cross-validation
    calls = re.findall(r"\$train_tool(.*?)2>>\s*\$log", text, flags=re.S)
    assert len(calls) == 3, len(calls)
    opts = [sorted(set(re.findall(r"--([a-z0-9-]+)=", c))) for c in calls]
    awks = sorted(set(re.findall(r"awk '(/FRAME_ACCURACY/[^']*)'", text)))
    assert len(awks) == 1, awks
    return {"source": REL + ":76-120", "prerun_options": opts[0], "train_options": opts[1], "crossval_options": opts[2],
            "awk_program": awks[0], "positional": {"train": ["feats", "labels", "model-in", "model-out"],
                                                   "crossval": ["feats", "labels", "model-in"]}}


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scheduler_contract.json")
    json.dump(extract(root), open(out, "w"), indent=1, sort_keys=True)
    print(open(out).read())
