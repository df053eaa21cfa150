"""Avatar modules — the L3 drop-in boundary of HFA-GP.

Counterparts of /root/reference/code/networks/headnerf.py:
  HeadNeRF_final :44-134   HeadNeRF_3DMM :162-219   HeadNeRF_Audio :222-279
  Weights_3DMM   :138-158  AudioAttNet   :284-314   AudioNet       :319-349
Same class names, constructor signatures, method names (`get_weights`, `get_latent`, `get_image`,
`forward`, `get_delta`), parameter names (`bases`, `delta`, `bases_2`, `delta_2`, `encoder.*`,
`weights_3dmm.fc.N.*`, `generator.*`) and side effects (the IN-PLACE label flip of columns
[1,2,5,6,9,10] on the caller's tensor).  `self.generator` is the MI355X `TriPlaneGenerator`
(generator.py) instead of an unpickled EG3D network.

Latent-basis layer: Q = qr((bases + 1e-8)^T), ws = (alpha @ Q^T).view(B,14,512) + delta.  On the GPU the QR of
the [7168, K] panel is ops.TallSkinnyQR (Gram-matrix Householder kernel, LAPACK's signs); when `bases` does not require
grad the factor is cached (the reference re-factorises on every call, headnerf.py:91).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import nn

from .encoder3d import Encoder, EqualLinear
from .generator import load_G_official

FLIP_COLUMNS = [1, 2, 5, 6, 9, 10]      # headnerf.py:108,132,201,214,261,274
NUM_WS = 14                               # headnerf.py:55
_FLIP_SIGN = {}                           # (device, dtype) -> [25] tensor of +-1


def flip_label_(label: torch.Tensor) -> torch.Tensor:
    """`label[:, [1,2,5,6,9,10]] *= -1` IN PLACE on the caller's tensor (the side effect callers of the reference
    see), as one multiplication by a cached sign vector: no host-built index tensor, so it is also legal inside a
    HIP-graph capture."""
    key = (label.device, label.dtype, label.shape[-1])
    sign = _FLIP_SIGN.get(key)
    if sign is None:
        sign = torch.ones(label.shape[-1], dtype=label.dtype)
        sign[FLIP_COLUMNS] = -1
        sign = _FLIP_SIGN[key] = sign.to(label.device)
    return label.mul_(sign)


def load_bases(device, base_dir, dim_shape):
    """PTI pivot initialisation (headnerf.py:12-23): one `<dir>/0.pt` latent per basis vector."""
    bases = torch.randn(dim_shape, 18, 512)
    dirs = os.listdir(base_dir)
    for i in range(dim_shape):
        bases[i] = torch.load(base_dir + dirs[i] + "/0.pt").squeeze(0).detach()
    return bases.to(device)


def toogle_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


class _QrMonitor:
    """Host-side watch on the conditioning of the latent basis for the fast QR (see `_LatentBasis._qr`).  Holds a pinned
    buffer and a HIP event, so it is deliberately NOT copied or pickled with the module: a copy starts fresh."""

    # orthogonality defect of the fast QR's first pass above which the instance switches to the library Householder QR
    # for good: the re-orthogonalised result is O(eps)-orthonormal up to a defect of ~0.3 (cond(A) ~ 3000 in fp32), the
    # monitor reads the status word of the PREVIOUS call (no host sync in the step), so the margin is wide
    DEFECT_LIMIT = 1e-2

    def __init__(self):
        self.fallback = False
        self.checked = False
        self._pending = None          # (pinned host tensor, event) of the status word in flight
        self._host = None

    def __deepcopy__(self, memo):
        return _QrMonitor()

    def __reduce__(self):
        return (_QrMonitor, ())

    def bad(self, st) -> bool:
        defect, broke = st
        return broke != 0.0 or not (defect <= self.DEFECT_LIMIT)

    def watch(self, status: torch.Tensor) -> None:
        if self._pending is not None:         # the previous status word is still in flight
            return
        if self._host is None:
            self._host = torch.empty(2, dtype=torch.float32, pin_memory=True)
        self._host.copy_(status, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending = (self._host, ev)

    def poll(self) -> None:
        if self._pending is not None and self._pending[1].query():
            host, _ = self._pending
            self._pending = None
            if self.bad(host.tolist()):
                import warnings
                warnings.warn("HeadNeRF latent basis became ill-conditioned for the Gram-matrix QR "
                              f"(first-pass orthogonality defect {host[0].item():.2e}); switching to torch.linalg.qr")
                self.fallback = True


class _LatentBasis(nn.Module):
    """Shared implementation of the K-dimensional W+ subspace (rows A1-A3 of SURVEY.md §8a)."""

    def _init_basis(self, device, dim, dim_shape):
        self.dim, self.dim_shape = dim, dim_shape
        bases = torch.randn(dim_shape, NUM_WS * dim).to(device)
        self.bases = nn.Parameter(bases, requires_grad=True)
        self.delta = nn.Parameter(bases.mean(dim=0), requires_grad=True)
        self._q_cache = None

    def _select(self, person_2: bool):
        return self.bases, self.delta

    def _orthonormal(self, bases: torch.Tensor) -> torch.Tensor:
        if not bases.requires_grad:
            # a FROZEN basis (reenactment loops): Q is cached.  A trainable one is never cached, also under no_grad (`Trainer.sample`
            # between steps): the key is the parameter's `_version`, which torch's fused Adam does not advance (round 5, see
            # TriPlaneGenerator._refresh_tuned) — the cached Q would be the basis of the first sample call for ever.
            key = (id(bases), bases._version, bases.data_ptr())
            if self._q_cache is not None and self._q_cache[0] == key:
                return self._q_cache[1]
            q = self._qr((bases.detach() + 1e-8).T)
            self._q_cache = (key, q)
            return q
        self._q_cache = None          # (trained, then frozen on the same object: the frozen branch must start from an empty cache)
        if not torch.is_grad_enabled():
            return self._qr((bases.detach() + 1e-8).T)
        return self._qr((bases + 1e-8).T)

    def _qr(self, a: torch.Tensor) -> torch.Tensor:
        """Q of torch.linalg.qr(a, 'reduced') (headnerf.py:85-91).  On the GPU the [7168, K<=64] panel goes through the
        Gram-matrix Householder kernel + one Cholesky re-orthogonalisation (ops.TallSkinnyQR: LAPACK's signs, ~0.3 ms
        instead of ~1.4 ms of rocSOLVER launches; it works on A^T A, so it is only used for tall panels, m >= 8 K, and
        only while the conditioning monitor is green); elsewhere (CPU, K > 64, ill-conditioned basis such as a
        PTI-pivot initialisation with a large shared mean) through torch.linalg.qr.
        Conditioning monitor (`_QrMonitor`): the first call on an instance checks the defect of pass 1 synchronously
        (one host sync, once); later calls poll the status word of the previous call through a pinned host copy + event,
        so a basis that degrades during training switches to the library path one step late at worst, without a sync
        per step."""
        mon = self.__dict__.setdefault("_qr_monitor", _QrMonitor())
        fast = (a.is_cuda and a.dtype == torch.float32 and a.shape[1] <= 64 and a.shape[0] >= 8 * a.shape[1]
                and not mon.fallback)
        if not fast:
            return torch.linalg.qr(a, mode="reduced")[0]
        from . import ops
        mon.poll()
        status = torch.empty(2, device=a.device, dtype=torch.float32)
        q = ops.TallSkinnyQR.apply(a, status)
        if not mon.checked:
            mon.checked = True
            if mon.bad(status.tolist()):
                mon.fallback = True
                return torch.linalg.qr(a, mode="reduced")[0]
            return q
        mon.watch(status)
        return q

    @property
    def _qr_fallback(self) -> bool:
        return self.__dict__.get("_qr_monitor") is not None and self.__dict__["_qr_monitor"].fallback

    def get_latent(self, weights: Optional[torch.Tensor], person_2: bool = False):
        bases, delta = self._select(person_2)
        q = self._orthonormal(bases)                     # [14*dim, K]
        if weights is None:
            return q
        out = weights @ q.T                              # == sum_k diag(alpha) Q^T  (headnerf.py:96-98)
        return out.view(weights.shape[0], -1, self.dim) + delta.view(-1, self.dim)

    def get_image(self, latent: torch.Tensor, label: torch.Tensor, **renderer_uniforms) -> torch.Tensor:
        flip_label_(label)                               # in place, on the caller's tensor
        return self._synthesis(latent, label, renderer_uniforms)

    def _synthesis(self, latent, label, renderer_uniforms) -> torch.Tensor:
        """`generator.synthesis(latent, c=label, noise_mode='const')['image']` (headnerf.py:112).  `renderer_uniforms`
        (keyword-only `u_strat` [B,R,Sc], `u_imp` [B*R,Sf]) is a TEST HOOK: EG3D's renderer draws its stratified-jitter and
        importance-sampling uniforms inside the call, which makes two renders of one frame differ; a parity
        test hands the same draws to both sides.  Absent (every reference call site), the generator draws them itself."""
        bad = set(renderer_uniforms) - {"u_strat", "u_imp"}
        if bad:
            raise TypeError(f"unexpected keyword arguments {sorted(bad)}")
        kw = {k: v for k, v in renderer_uniforms.items() if v is not None}
        return self.generator.synthesis(latent, c=label, noise_mode="const", **kw)["image"]


class HeadNeRF_final(_LatentBasis):
    """RGB-driven avatar (headnerf.py:44-134)."""

    def __init__(self, args, size, device, dim=512, dim_shape=20, run_id="nerface2",
                 emb_dir="./PTI/embeddings/", use_softmax=False):
        super().__init__()
        self.base_dir = emb_dir + run_id
        self.device = device
        self.encoder = Encoder(size, dim, dim_shape, use_softmax, args.out_pose)
        self.args = args
        self.out_pose = args.out_pose
        self._init_basis(device, dim, dim_shape)
        if getattr(args, "person_2", False):
            if getattr(args, "init", False):
                bases_2 = load_bases(device, emb_dir + args.run_id_2 + "/PTI/", dim_shape).view(dim_shape, -1)
            else:
                bases_2 = torch.randn(dim_shape, NUM_WS * dim).to(device)
            if not args.same_bases:
                self.bases_2 = nn.Parameter(bases_2, requires_grad=True)
            self.delta_2 = nn.Parameter(bases_2.mean(dim=0), requires_grad=True)
        self.generator = load_G_official(args, device)

    def _select(self, person_2: bool):
        if not person_2:
            return self.bases, self.delta
        return (self.bases if self.args.same_bases else self.bases_2), self.delta_2

    def get_delta(self, person_2=False):
        return (self.delta_2 if person_2 else self.delta).view(-1, self.dim)

    def get_weights(self, image):
        return self.encoder(image)          # (weights, pose) when out_pose

    def forward(self, image, label, person_2=False, **renderer_uniforms):
        flip_label_(label)
        if self.out_pose:
            weights, pose = self.encoder(image)
        else:
            weights, pose = self.encoder(image), None
        latent = self.get_latent(weights, person_2)
        img = self._synthesis(latent, label, renderer_uniforms)
        return (img, pose) if self.out_pose else img


class Weights_3DMM(nn.Module):
    """Seven activation-less EqualLinear layers: expression coefficients → basis coordinates (:138-158)."""

    def __init__(self, input_dim=76, dim=512, dim_shape=50, use_softmax=False):
        super().__init__()
        dims = [input_dim] + [dim] * 6 + [dim_shape]
        self.fc = nn.Sequential(*[EqualLinear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self.softmax = nn.Softmax(dim=1)
        self.use_softmax = use_softmax

    def forward(self, x):
        w = self.fc(x)
        return self.softmax(w) if self.use_softmax else w


class _ParamDriven(_LatentBasis):
    def __init__(self, args, size, device, dim=512, dim_shape=20, run_id="nerface2",
                 emb_dir="./PTI/embeddings/", use_softmax=False):
        super().__init__()
        self.base_dir = emb_dir + run_id
        self.device = device
        self.weights_3dmm = Weights_3DMM(input_dim=args.params_len, dim=dim, dim_shape=dim_shape,
                                         use_softmax=use_softmax)
        self._init_basis(device, dim, dim_shape)
        self.generator = load_G_official(args, device)

    def get_weights(self, params):
        return self.weights_3dmm(params)

    def forward(self, params, label, person_2=False, **renderer_uniforms):
        flip_label_(label)
        latent = self.get_latent(self.weights_3dmm(params), person_2)
        return self._synthesis(latent, label, renderer_uniforms)


class HeadNeRF_3DMM(_ParamDriven):
    """3DMM-expression-driven avatar (headnerf.py:162-219)."""


class HeadNeRF_Audio(_ParamDriven):
    """Audio-feature-driven avatar (headnerf.py:222-279)."""


def _conv1d_k3(x: torch.Tensor, conv: nn.Conv1d) -> torch.Tensor:
    """`conv(x)` for the kernel-3, padding-1 Conv1d layers of AudioNet (stride 2) / AudioAttNet (stride 1) as ONE matmul over
    the unfolded windows.  On the GPU nn.Conv1d goes through MIOpen, whose first call per shape benchmarks candidate kernels
    (seconds on a fresh box) and whose pick depends on its find mode; these layers are [N, <= 64, <= 16] tensors — a GEMM of a
    few kFLOP per window — so the MI355X path keeps them on rocBLAS.  Same parameters, same state_dict, same values up to
    fp32 summation order (CPU tensors take nn.Conv1d itself: bit-identical to the reference there)."""
    if not x.is_cuda:
        return conv(x)
    win = torch.nn.functional.pad(x, (1, 1)).unfold(2, 3, conv.stride[0])       # [N, C, L_out, 3]
    y = torch.einsum("nclk,ock->nol", win, conv.weight)
    return y + conv.bias[None, :, None] if conv.bias is not None else y


def _run_conv1d_stack(x: torch.Tensor, seq: nn.Sequential) -> torch.Tensor:
    for m in seq:
        x = _conv1d_k3(x, m) if isinstance(m, nn.Conv1d) else m(x)
    return x


class AudioAttNet(nn.Module):
    """Attention over a window of per-frame audio features (:284-314): scores from the first `dim_aud`
    dims, weighted sum over all dims."""

    def __init__(self, dim_aud=32, seq_len=8):
        super().__init__()
        self.seq_len, self.dim_aud = seq_len, dim_aud
        chans = [dim_aud, 16, 8, 4, 2, 1]
        layers = []
        for a, b in zip(chans[:-1], chans[1:]):
            layers += [nn.Conv1d(a, b, kernel_size=3, stride=1, padding=1, bias=True), nn.LeakyReLU(0.02, True)]
        self.attentionConvNet = nn.Sequential(*layers)
        self.attentionNet = nn.Sequential(nn.Linear(seq_len, seq_len, bias=True), nn.Softmax(dim=1))

    def forward(self, x):
        y = x[..., :self.dim_aud].permute(1, 0).unsqueeze(0)
        y = _run_conv1d_stack(y, self.attentionConvNet)
        y = self.attentionNet(y.view(1, self.seq_len)).view(self.seq_len, 1)
        return torch.sum(y * x, dim=0)

    def forward_windows(self, x):
        """Batched form for the reenactment harness: x [N, seq_len, D] (one smoothing window per frame) → [N, D];
        row n equals ``forward(x[n])``."""
        y = _run_conv1d_stack(x[..., :self.dim_aud].permute(0, 2, 1), self.attentionConvNet)          # [N, 1, seq_len]
        y = self.attentionNet(y.view(-1, self.seq_len))                             # softmax over the window
        return torch.sum(y.unsqueeze(-1) * x, dim=1)


class AudioNet(nn.Module):
    """DeepSpeech window [N,16,29] → per-frame audio feature [N, dim_aud] (:319-349)."""

    def __init__(self, dim_aud=76, win_size=16):
        super().__init__()
        self.win_size, self.dim_aud = win_size, dim_aud
        chans = [29, 32, 32, 64, 64]
        layers = []
        for a, b in zip(chans[:-1], chans[1:]):
            layers += [nn.Conv1d(a, b, kernel_size=3, stride=2, padding=1, bias=True), nn.LeakyReLU(0.02, True)]
        self.encoder_conv = nn.Sequential(*layers)
        self.encoder_fc1 = nn.Sequential(nn.Linear(64, 64), nn.LeakyReLU(0.02, True), nn.Linear(64, dim_aud))

    def forward(self, x):
        half = int(self.win_size / 2)
        x = x[:, 8 - half: 8 + half, :].permute(0, 2, 1)
        x = _run_conv1d_stack(x, self.encoder_conv).squeeze(-1)
        return self.encoder_fc1(x).squeeze()
