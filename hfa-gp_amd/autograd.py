"""Backward pass of ``TriPlaneGenerator.synthesis`` w.r.t. the latent ``ws`` (generator frozen) —
what HFA-GP's fitting step needs for its first ``tune_iter`` iterations
(/root/reference/code/trainer_rgb.py:59-60,73-98: ``g_loss.backward()`` flows through
``generator.synthesis`` into ``bases`` / ``delta`` / the driver net).

All arithmetic is HIP (ops.py); this file only walks the tape in reverse:

  image ── toRGB/skip adjoints ──► SR block1 ──► SR block0 ──► feature image ──► ray-march backward
        ──► d planes ──► backbone blocks 256 … 4 ──► per-layer style gradients ──► d ws

Per activation tensor X one fused ``pointwise_bwd`` pass sums the input gradients of X's consumers, reduces
their style gradients and pushes the result through the producer's clamp / leaky-ReLU / demodulation; the
GEMM-shaped adjoints (conv bwd-data) run on the forward MFMA kernel with transposed weights.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch

from . import ops


def _masked(g: torch.Tensor, y: Optional[torch.Tensor], clamp: Optional[float]) -> torch.Tensor:
    """Gradient through bias_act's clamp: zero where the (pre-clamp) output reached the limit."""
    if clamp is None or y is None:
        return g
    return g * (y.abs() < clamp)


class _Backward:
    def __init__(self, gen, tape, d_ws: torch.Tensor, ws: torch.Tensor, param_grads: bool):
        self.gen, self.tape, self.d_ws, self.ws, self.pg = gen, tape, d_ws, ws, param_grads
        self.grads = {}            # id(parameter) -> gradient (only when the generator is being tuned)
        self.by_id = {}            # id(parameter) -> parameter
        self.released = set()      # ids already handed to the gradient sink / accumulated in place
        # in-place mode (set by the trainer's step scope): this pass ADDS every generator gradient into the parameter's
        # existing .grad (a slice of the trainer's flat buffer) itself — the conv weight gradients straight from the reducer
        # kernel, everything else in one multi-tensor add per release — and autograd receives None for them (round 5: the
        # tuned step spent ~140 launches of 2-3 us + 370 MB of traffic on AccumulateGrad's per-tensor adds)
        self.inplace = bool(getattr(gen, "_grad_inplace", False))
        self.pending = []          # style gradients of the whole pass: (item for ops.style_bwd_batch, affine module)
        self.deferred = []         # (partial, sums) of the fused passes whose reduction waits for flush_styles (generator frozen)
        self.direct_ready = []     # parameters whose .grad a kernel has accumulated into since the last release
        self.small = []            # in-place mode: (sums, bias.grad, noise_strength.grad) of the block's layers, one launch per block

    # bench.py's roofline_train: HIP events around the three kernel families that carry the backward pass (gen.timing keys
    # "bwd_data" [algorithmic flops 2 M N K of the adjoint conv], "pointwise_bwd" [bytes of the full-size tensors read +
    # written], "raymarch_bwd" [frames]); plain calls when timing is off
    def bwd_data(self, g: torch.Tensor, weight: torch.Tensor, cin: int, mode: int) -> torch.Tensor:
        if self.gen.timing is None:
            return ops.modconv(g, self.wt_t(weight), cin, mode)
        if mode == ops.CONVS2_BWD:                 # g = parity images [2,2,B,H+1,W+1,Cout] of the y_t gradient, output B x H x W
            n_pos = g.shape[2] * (g.shape[3] - 1) * (g.shape[4] - 1)
        else:
            n_pos = g.shape[0] * g.shape[1] * g.shape[2]
        taps = weight.shape[2] * weight.shape[3]
        flops = 2.0 * n_pos * weight.shape[0] * weight.shape[1] * taps
        key = {ops.CONV3X3_BWD: "bwd_data", ops.CONVS2_BWD: "bwd_data_up", ops.CONV1X1: "bwd_data_1x1"}[mode]
        return self.gen._timed(key, flops, ops.modconv, g, self.wt_t(weight), cin, mode)

    def pointwise(self, x: torch.Tensor, **kw):
        if not self.pg:                     # generator frozen: the sums are needed only by flush_styles — reduced there, all at once
            kw["deferred"] = self.deferred
        if self.gen.timing is None:
            return ops.pointwise_bwd(x, **kw)
        nbytes = 4.0 * (2 * x.numel() + sum(kw[k].numel() for k in ("dxs_conv", "dxs_rgb", "g_direct", "g_nchw3_a", "g_nchw3_b")
                                            if kw.get(k) is not None))
        return self.gen._timed("pointwise_bwd", nbytes, ops.pointwise_bwd, x, **kw)

    def wgrad(self, x: torch.Tensor, styles, g: torch.Tensor, weight: torch.Tensor, mode: int, **kw) -> None:
        """Weight-gradient GEMM of one layer (generator being tuned) into the layer's gradient; timing keys "wgrad" (3x3),
        "wgrad_up" (the four parity images of the up-sampling conv), "wgrad_1x1" (96-channel toRGB): algorithmic flops
        2 * positions * Cin * Cout * taps, the split-K reducer that follows the GEMM included."""
        direct = self.inplace and weight.requires_grad and weight.grad is not None and id(weight) not in self.grads
        if direct:
            kw["out"] = weight.grad            # the reducer accumulates into the flat-buffer slice
        if self.gen.timing is None:
            dw = ops.conv_wgrad(x, styles, g, weight.detach(), mode, **kw)
        else:
            flops = 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * weight.numel()
            key = {ops.CONV3X3: "wgrad", ops.CONVT3X3_UP2: "wgrad_up", ops.CONV1X1: "wgrad_1x1"}[mode]
            dw = self.gen._timed(key, flops, ops.conv_wgrad, x, styles, g, weight.detach(), mode, **kw)
        if direct:
            self.released.add(id(weight))
            sink = getattr(self.gen, "_grad_sink", None)
            if sink is not None:
                sink(weight, None)             # (already accumulated: only counted as ready)
        else:
            self._acc(weight, dw)

    def _direct(self, param: torch.Tensor) -> bool:
        """May a kernel accumulate into param.grad itself (the trainer's step scope: .grad is a zeroed slice of the flat buffer)?"""
        return self.inplace and param.requires_grad and param.grad is not None and param.grad.is_contiguous() \
            and id(param) not in self.grads and id(param) not in self.released

    def _mark_direct(self, param: torch.Tensor):
        self.released.add(id(param))
        self.direct_ready.append(param)

    def _acc(self, param: torch.Tensor, g: torch.Tensor):
        key = id(param)
        self.grads[key] = g if key not in self.grads else self.grads[key] + g
        self.by_id[key] = param

    def release_ready(self):
        """Hand the parameter gradients computed so far to the generator's gradient sink (a multi-GPU trainer's
        bucketed all-reduce, trainer.BucketedAllReduce): each parameter is accumulated exactly once per pass, so
        whatever is in `grads` is final.  The sink adds the gradient into the parameter's .grad slice and may start
        the collective of a finished bucket while the rest of the backward pass is still being enqueued; released
        parameters get None from autograd.  Without a sink nothing happens and autograd receives every gradient."""
        sink = getattr(self.gen, "_grad_sink", None)
        if self.small:
            ops.bias_noise_grads(self.small)
            self.small = []
        if self.direct_ready:
            if sink is not None:
                for p in self.direct_ready:
                    sink(p, None)          # (already accumulated: only counted as ready)
            self.direct_ready = []
        if self.inplace:
            ready = [k for k in self.grads if self.by_id[k].requires_grad and self.by_id[k].grad is not None]
            if ready:
                prms = [self.by_id[k] for k in ready]
                torch._foreach_add_([p.grad for p in prms], [self.grads.pop(k).view_as(p.grad) for k, p in zip(ready, prms)])
                self.released.update(ready)
                if sink is not None:
                    for p in prms:
                        sink(p, None)
            return
        if sink is None:
            return
        for key in list(self.grads):
            prm = self.by_id[key]
            if prm.requires_grad and prm.grad is not None:
                sink(prm, self.grads.pop(key))
                self.released.add(key)

    def affine_grads(self, affine, dstot: torch.Tensor, row: int, dA: Optional[torch.Tensor] = None,
                     db: Optional[torch.Tensor] = None):
        """dA / db: zero-initialised views of one flat buffer (flush_styles: one fill for all layers instead of two each)."""
        dA = torch.zeros_like(affine.weight) if dA is None else dA
        db = torch.zeros_like(affine.bias) if db is None else db
        ops.affine_grad(dstot, self.ws[:, row], dA, db)
        self._acc(affine.weight, dA)
        self._acc(affine.bias, db)

    # transposed-weight images for the bwd-data GEMMs (cached on the generator like the forward ones)
    def wt_t(self, weight: torch.Tensor) -> torch.Tensor:
        return self.gen._gemm_image(weight, transposed=True)

    def style_grad(self, rec_layer: dict, ds: torch.Tensor, dd: Optional[torch.Tensor], style_gain: float = 1.0,
                   affine=None, wsq=None, dcoef=None):
        affine = affine if affine is not None else rec_layer["layer"].affine
        ops.style_bwd(ds, dd, rec_layer["styles"], dcoef, wsq, affine.weight, self.d_ws[:, rec_layer["row"]],
                      style_gain, accumulate=True)

    # ------------------------------------------------------------------ one SynthesisBlock, in reverse
    def image_chain(self, recs: List[dict], g_img: torch.Tensor, side) -> List[tuple]:
        """The image-gradient side chain of the backbone, run AHEAD on the second stream (generator.side_stream): per block
        (last first) the adjoint of `upsample2d(img)` and the toRGB adjoint `dxs_rgb` depend on the block's image gradient
        only — never on the conv chain — so all of them are enqueued here while the launch stream walks the blocks; `block`
        waits for its entry's event.  Returns [(g_img_prev, g_y, dxs_rgb, event)] in walking order."""
        main = torch.cuda.current_stream(g_img.device)
        for rec in recs:                          # transposed weight images are cached on the generator: build them on the
            self.wt_t(rec["rgb"]["torgb"].weight)  # launch stream, where later steps will also find them
        side.wait_stream(main)
        g_img.record_stream(side)
        out = []
        with torch.cuda.stream(side):
            g = g_img
            for rec in recs:
                rgb = rec["rgb"]
                tr = rgb["torgb"]
                g_prev = ops.upsample2d_bwd(g, channels_last=True) if rec["img_in"] is not None else None
                g_y = _masked(g, rgb["y"], rgb["clamp"]).contiguous()
                dxs_rgb = self.bwd_data(g_y, tr.weight, tr.weight.shape[1], ops.CONV1X1)
                ev = torch.cuda.Event()
                ev.record(side)
                out.append((g_prev, g_y, dxs_rgb, ev))
                g = g_prev
        return out

    def block(self, rec: dict, g_img, dxs_next: Optional[torch.Tensor], s_next: Optional[torch.Tensor],
              next_layer_rec: Optional[dict], ahead: Optional[tuple] = None):
        """g_img: gradient of the block's output skip image (NCHW small / NHWC 96-ch).
        dxs_next / s_next: raw bwd-data of the NEXT block's conv0 w.r.t. (x*s) and its styles (or None).
        Returns (dxs of this block's conv0 w.r.t. its modulated input | None, conv0 rec | None, g_img_prev)."""
        rgb, c1, c0 = rec["rgb"], rec["conv1"], rec["conv0"]
        tr = rgb["torgb"]
        cin = tr.weight.shape[1]
        x1 = c1["out"]
        # ---- toRGB + skip
        g_img_prev = None
        if ahead is not None:
            # computed ahead on the side stream (image_chain): join here, where this block's fused pass consumes it
            g_img_prev, g_y, dxs_rgb, ev = ahead
            main = torch.cuda.current_stream(x1.device)
            main.wait_event(ev)
            for t in (g_y, dxs_rgb):
                t.record_stream(main)
            kw = dict(dxs_rgb=dxs_rgb, s_rgb=rgb["styles"])
        elif rec["img_in"] is not None:
            g_img_prev = ops.upsample2d_bwd(g_img, channels_last=not rgb["small"])
        if ahead is not None:
            pass
        elif rgb["small"]:
            if self.pg:          # (the bias gradient below sums the masked gradient itself)
                g_y = _masked(g_img, rgb["y_pre"], rgb["clamp"]).contiguous()
                kw = dict(g_rgb_small=g_y)
            else:                # generator frozen: the fused pass applies the clamp mask [|y_pre| < clamp] while it reads
                kw = dict(g_rgb_small=g_img.contiguous(), y_rgb_small=rgb["y_pre"], clamp_rgb_small=rgb["clamp"])
            kw.update(w_rgb_small=tr.weight.detach().reshape(tr.weight.shape[0], cin), s_small=rgb["styles"])
        else:
            g_y = _masked(g_img, rgb["y"], rgb["clamp"]).contiguous()
            dxs_rgb = self.bwd_data(g_y, tr.weight, cin, ops.CONV1X1)
            kw = dict(dxs_rgb=dxs_rgb, s_rgb=rgb["styles"])
        # ---- X = conv1 output: consumers = next conv0 (+) toRGB; producer = conv1
        g_conv1, sums = self.pointwise(x1, dxs_conv=dxs_next, s_conv=s_next, producer=c1["producer"],
                                       param_grads=self.pg, **kw)
        if next_layer_rec is not None:
            next_layer_rec["ds"] = sums[:, 0]
        ds_rgb = sums[:, 2] if rgb["small"] else sums[:, 1]
        self.pending.append(((ds_rgb, None, rgb["styles"], None, None, tr.affine.weight, rgb["row"],
                              1.0 / math.sqrt(cin)), tr.affine))
        c1["dd"] = sums[:, 3]
        if self.pg:
            if rgb["small"]:
                co = tr.weight.shape[0]
                dw = (sums[:, 6:6 + co] * rgb["styles"][:, None, :]).sum(0)          # [Co, C]: tiny host-side glue
                self._acc(tr.weight, dw.reshape(tr.weight.shape))
                g_cl = g_y.permute(0, 2, 3, 1).contiguous()
            else:
                self.wgrad(x1, rgb["styles"], g_y, tr.weight, ops.CONV1X1)
                g_cl = g_y
            if self._direct(tr.bias):                  # straight into the .grad slice (no zero-filled temporary, no add)
                ops.channel_sum(g_cl, tr.bias.grad)
                self._mark_direct(tr.bias)
            else:
                db = torch.zeros_like(tr.bias)
                ops.channel_sum(g_cl, db)
                self._acc(tr.bias, db)
            self.layer_param_grads(c1, sums, g_conv1)
        # ---- conv1 bwd-data
        c1_cin = c1["layer"].weight.shape[1]
        dxs1 = self.bwd_data(g_conv1, c1["layer"].weight, c1_cin, ops.CONV3X3_BWD)
        if c0 is None:
            # b4: the input is the learned constant (broadcast over the batch): only the style gradient is needed
            xc = c1["x"].expand(dxs1.shape[0], -1, -1, -1).contiguous()
            gconst, s0 = self.pointwise(xc, dxs_conv=dxs1, s_conv=c1["styles"])
            c1["ds"] = s0[:, 0]
            self.finish_layer(c1)
            if self.pg:
                self._acc(rec["const"], gconst.sum(0).permute(2, 0, 1).contiguous())
            return None, None, g_img_prev
        # ---- X = conv0 output: consumer = conv1; producer = conv0 (up-sampling layer)
        g_conv0, s0 = self.pointwise(c0["out"], dxs_conv=dxs1, s_conv=c1["styles"], producer=c0["producer"],
                                     param_grads=self.pg)
        c1["ds"] = s0[:, 0]
        c0["dd"] = s0[:, 3]
        self.finish_layer(c1)
        gph = ops.upfir_bwd(g_conv0)
        if self.pg:
            self.layer_param_grads(c0, s0, gph)
        c0_cin = c0["layer"].weight.shape[1]
        dxs0 = self.bwd_data(gph, c0["layer"].weight, c0_cin, ops.CONVS2_BWD)
        return dxs0, c0, g_img_prev

    def layer_param_grads(self, rec: dict, sums_out: torch.Tensor, g: torch.Tensor):
        """weight / bias / noise-strength gradients of one SynthesisLayer.  `sums_out` = reductions of the pass over
        the layer's OUTPUT; `g` = gradient w.r.t. its raw conv output (parity images for the up-sampling layer)."""
        layer = rec["layer"]
        x = rec["x"]
        if x.shape[0] != g.shape[-4]:                      # b4: the learned constant is shared by the batch
            x = x.expand(g.shape[-4], -1, -1, -1).contiguous()
        mode = ops.CONVT3X3_UP2 if rec["up"] == 2 else ops.CONV3X3
        # gradient GEMMs follow the generator's precision class: exact fp32 MFMA when conv_precision is "fp32", else
        # split-bf16 (the 3x3 layers with 64-multiple channels; bf16 parts keep a gradient's exponent range)
        wprec = "fp32" if self.gen.conv_precision == "fp32" else "bf16x3"
        self.wgrad(x, rec["styles"], g, layer.weight, mode, dd=rec["dd"], dcoef=rec["dcoef"], precision=wprec)
        noise = layer.noise_strength if rec["producer"]["noise"] is not None else None
        if self._direct(layer.bias) and (noise is None or self._direct(noise)):
            # straight into the .grad slices, all layers of the block in one launch (release_ready): per layer the framework
            # ran two reductions and the release two adds
            self.small.append((sums_out, layer.bias.grad, None if noise is None else noise.grad.view(1)))
            self._mark_direct(layer.bias)
            if noise is not None:
                self._mark_direct(noise)
            return
        self._acc(layer.bias, sums_out[:, 4].sum(0))
        if noise is not None:
            self._acc(noise, sums_out[:, 5].sum())

    def finish_layer(self, rec: dict):
        """ds (from the pass over the layer's input) and dd (from the pass over its output) are both known: queue the
        layer's style gradient (all of them run in two launches at the end of the pass, `flush_styles`)."""
        self.pending.append(((rec["ds"], rec["dd"], rec["styles"], rec["dcoef"], rec["wsq"], rec["layer"].affine.weight,
                              rec["row"], 1.0), rec["layer"].affine))

    def flush_styles(self):
        if self.deferred:
            ops.reduce_partials_batch(self.deferred)
        dstots = ops.style_bwd_batch([it for it, _ in self.pending], self.d_ws)
        if self.pg and all(self._direct(affine.weight) and self._direct(affine.bias) for _, affine in self.pending) \
                and len({id(affine) for _, affine in self.pending}) == len(self.pending):
            # every affine layer of the pass in one launch, straight into the .grad slices (was: a zero-filled staging buffer,
            # 26 launches, a multi-tensor add)
            ops.affine_grad_batch([(dstot, self.ws[:, it[6]], affine.weight.grad, affine.bias.grad)
                                   for (it, affine), dstot in zip(self.pending, dstots)])
            for _, affine in self.pending:
                self._mark_direct(affine.weight)
                self._mark_direct(affine.bias)
        elif self.pg:
            sizes = [(affine.weight.numel(), affine.bias.numel()) for _, affine in self.pending]
            flat = torch.zeros(sum(a + b for a, b in sizes), device=self.d_ws.device, dtype=torch.float32)
            off = 0
            for ((it, affine), dstot), (na, nb) in zip(zip(self.pending, dstots), sizes):
                dA = flat[off: off + na].view_as(affine.weight)
                db = flat[off + na: off + na + nb].view_as(affine.bias)
                off += na + nb
                self.affine_grads(affine, dstot, it[6], dA, db)
        self.pending = []


class SynthesisFn(torch.autograd.Function):
    """inputs: ws, c, u_strat, u_imp, gen, *generator parameters (listed only so that autograd can hand their
    gradients back when the generator is being tuned; the forward reads them from `gen`)."""

    @staticmethod
    def forward(ctx, ws, c, u_strat, u_imp, gen, *params):
        tape = {}
        ws_c = ws.detach().float().contiguous()
        with torch.no_grad():
            img, rgb_raw, depth, planes, feat_img = gen._forward_impl(ws_c, c.detach().float().contiguous(), u_strat,
                                                                      u_imp, tape)
        gen._last_extras = (planes, feat_img)        # for synthesis(return_planes=True)
        ctx.gen, ctx.tape, ctx.ws_c = gen, tape, ws_c
        ctx.params = params
        ctx.pg = any(p.requires_grad for p in params)
        ctx.ws_shape = ws.shape
        ctx.mark_non_differentiable(depth)
        return img, rgb_raw, depth

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g_img, g_raw, _g_depth):
        gen, tape = ctx.gen, ctx.tape
        if tape is None:
            raise RuntimeError("TriPlaneGenerator.synthesis: backward called a second time — the saved activations "
                               "(several GB at 512^2) are released by the first backward pass; sum the losses before "
                               "calling backward instead of backpropagating them one by one (retain_graph is not supported)")
        cfg = gen.cfg
        b = tape["batch"]
        dev = tape["planes"].device
        d_ws = torch.zeros(ctx.ws_shape, device=dev, dtype=torch.float32)
        bw = _Backward(gen, tape, d_ws, ctx.ws_c, ctx.pg)
        if g_img is None:
            g_img = torch.zeros(b, cfg.img_channels, cfg.img_resolution, cfg.img_resolution, device=dev)
        g_img = g_img.float().contiguous()

        # ---- super-resolution, block1 then block0
        sr1, sr0 = tape["sr"][1], tape["sr"][0]
        dxs, c0rec, g_rgb = bw.block(sr1, g_img, None, None, None)
        bw.release_ready()
        dxs, c0rec0, g_rgb_raw = bw.block(sr0, g_rgb, dxs, c0rec["styles"], c0rec)
        bw.release_ready()
        bw.finish_layer(c0rec)
        # ---- feature image: consumer = SR block0.conv0, plus the first 3 channels through image_raw
        feat_img = tape["feat_img"]
        # (image_raw = channels 0..2 of the feature image: the two NCHW gradients are added inside the fused pass)
        g_feat, s = bw.pointwise(feat_img.contiguous(), dxs_conv=dxs, s_conv=c0rec0["styles"],
                                 g_nchw3_a=g_rgb_raw.contiguous(), g_nchw3_b=None if g_raw is None else g_raw.float().contiguous())
        c0rec0["ds"] = s[:, 0]
        bw.finish_layer(c0rec0)
        # ---- renderer
        res = cfg.neural_rendering_resolution
        net = gen.decoder.net
        dec_prm = (net["0"].weight, net["0"].bias, net["2"].weight, net["2"].bias)
        # (the kernel's ~9 M end-of-kernel atomics landing in the .grad slices themselves — `dec_out` — instead of four fresh zeroed
        # tensors was measured: the pass takes 1.17 - 1.20 ms instead of 0.98 - 0.99, and still does with the atomics cut to one set
        # per workgroup (an LDS reduction over its eight waves: no change either way) — it is WHERE they land, not how many.
        # HFAGP_DEV_DEC_DIRECT=1 re-enables it for A/B timing)
        dec_direct = ctx.pg and os.environ.get("HFAGP_DEV_DEC_DIRECT", "0") == "1" and all(bw._direct(p) for p in dec_prm)
        rb = gen._timed("raymarch_bwd", float(b), ops.raymarch_bwd, g_feat.view(b, res * res, 32), tape["planes"],
                        u_strat=tape["u_strat"], u_imp=tape["u_imp"], decoder_grads=ctx.pg,
                        planes_absmax=tape.get("planes_absmax"), state=tape.get("ray_state"),
                        dec_out=tuple(p.grad for p in dec_prm) if dec_direct else None,
                        **gen._render_args(tape["c"]))
        if ctx.pg:
            d_planes, dec = rb
            for prm, g in zip(dec_prm, dec):
                if dec_direct:
                    bw._mark_direct(prm)
                else:
                    bw._acc(prm, g)
            bw.release_ready()
        else:
            d_planes = rb
        # ---- backbone, last block first
        g_img_b = ops.planes_to_nhwc(d_planes)
        dxs, nxt = None, None
        walk = list(reversed(tape["backbone"]))
        side = gen.side_stream(b, dev) if all(not r["rgb"]["small"] for r in walk) else None
        ahead = bw.image_chain(walk, g_img_b, side) if side is not None else None
        for k, rec in enumerate(walk):
            s_next = nxt["styles"] if nxt is not None else None
            dxs_new, c0, g_img_b = bw.block(rec, g_img_b, dxs, s_next, nxt, ahead[k] if ahead is not None else None)
            if nxt is not None:
                bw.finish_layer(nxt)
            dxs, nxt = dxs_new, c0
            bw.release_ready()
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)     # (nothing of this pass is left on the side stream)
        bw.flush_styles()
        bw.release_ready()
        ctx.tape = None
        pgrads = tuple(bw.grads.get(id(p)) if p.requires_grad else None for p in ctx.params)
        return (d_ws, None, None, None, None) + pgrads
