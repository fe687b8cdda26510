# where does a tool spend its time per utterance?  (a diagnostic; run on the GPU box)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, 'tools')
import soak
rs = np.random.RandomState(1)
soak.write_model('/tmp/m.nnet', [40, 256, 256, 120], rs)
print(soak.write_tables('/tmp/t', 2000, 40, 120, rs))
PY
time (cat /tmp/t.feats.ark > /dev/null)
echo "--- forward, default"; time (kaldi_amd/bin/nnet-forward /tmp/m.nnet ark:/tmp/t.feats.ark ark:/dev/null 2>&1 | grep FINISHED)
echo "--- forward, --batch-frames=0"; time (kaldi_amd/bin/nnet-forward --batch-frames=0 /tmp/m.nnet ark:/tmp/t.feats.ark ark:/dev/null 2>&1 | grep FINISHED)
echo "--- forward, --profile"; kaldi_amd/bin/nnet-forward --profile=true /tmp/m.nnet ark:/tmp/t.feats.ark ark:/dev/null 2>&1 | tail -25
echo "--- forward, output to a file"; time (kaldi_amd/bin/nnet-forward /tmp/m.nnet ark:/tmp/t.feats.ark ark:/tmp/o.ark 2>&1 | grep FINISHED)
echo "--- reference copy-feats-like read: python"; time (python -c "
d=open('/tmp/t.feats.ark','rb').read(); print(len(d))")
