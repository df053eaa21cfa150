cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "qr or QR or tall_gram or latent or own_size or basis" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -12
for reg in "3dmm" "rgb"; do
  python tools/dev/bench_train.py 2 20 $reg 2>&1 | tail -1
done
