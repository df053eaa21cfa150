"""One variant of the lanes-48-63 reproducer (driven by tests/micro/lanes48/repro.sh; needs an MI355X).
HFAGP_LIB_PATH selects the library build.  Runs hfagp_torgb_skip_fwd N times on fixed inputs and counts the elements that
differ from the two-pass reference (hfagp_modconv_fwd 1x1 + hfagp_skip_upsample_add), which is bit-identical by construction;
for every differing element prints what identifies the hardware lane: the column inside its 32-position wave tile (MFMA row =
lane & 31) and the channel inside its 32-channel MFMA tile (register r and lane >> 5 pick the row group)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from hfa_gp_amd import ops  # noqa: E402

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
b, h, w, cin, cout = 6, 128, 128, 256, 96
x = (torch.randn(b, h, w, cin, generator=g) * 4.0).to(dev)
wgt = torch.randn(cout, cin, 1, 1, generator=g).to(dev)
s = (torch.randn(b, cin, generator=g) / math.sqrt(cin)).to(dev)
bias = torch.randn(cout, generator=g).to(dev)
img = torch.randn(b, h // 2, w // 2, cout, generator=g).to(dev)
wb = ops.weight_prep_prec(wgt, "f16x3")
y = ops.modconv(x, wb, cout, ops.CONV1X1, styles=s, bias=bias, act="linear", gain=1.0, ksplit=1)
ref = ops.skip_upsample_add(img, y)
bad_runs = bad_elems = 0
cols, chans = {}, {}
for _ in range(n_runs):
    out = ops.torgb_skip(x, wb, cout, s, bias, img)
    d = (out != ref)
    k = int(d.sum())
    if k:
        bad_runs += 1
        bad_elems += k
        idx = d.nonzero()
        for xcol, ch in zip((idx[:, 2] % 32).tolist(), (idx[:, 3] % 32).tolist()):
            cols[xcol] = cols.get(xcol, 0) + 1
            chans[ch] = chans.get(ch, 0) + 1
print(f"variant {os.environ.get('HFAGP_VARIANT', '?')}: {n_runs} runs x {ref.numel()} outputs: {bad_runs} runs with differences, "
      f"{bad_elems} differing elements; column-in-tile histogram {dict(sorted(cols.items()))}; channel-in-tile {dict(sorted(chans.items()))}")
