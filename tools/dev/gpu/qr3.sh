cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "qr or QR or tall_gram" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -12
bash tools/dev/gpu/qr2.sh 2>&1 | grep "tall_gram\|qr_"
