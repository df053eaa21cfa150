"""Developer timing of the RGB driver (Encoder(256), PyTorch-ROCm / MIOpen) forward + backward at the fitting batch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.encoder3d import Encoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")


def run(tag, enc, x, iters=20):
    for _ in range(3):
        enc(x).sum().backward()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        enc(x).sum().backward()
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t) / iters * 1e3:.2f} ms fwd+bwd (B={B})")


torch.manual_seed(0)
enc = Encoder(256, 512, 50).to(dev)
x = torch.randn(B, 3, 256, 256, device=dev)
run("default", enc, x)
torch.backends.cudnn.benchmark = True
run("cudnn.benchmark", enc, x)
enc_cl = enc.to(memory_format=torch.channels_last)
run("channels_last + benchmark", enc_cl, x.contiguous(memory_format=torch.channels_last))
with torch.autocast("cuda", dtype=torch.bfloat16):
    run("bf16 autocast + channels_last", enc_cl, x.contiguous(memory_format=torch.channels_last))
