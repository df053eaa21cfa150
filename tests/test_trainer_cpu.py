"""Host logic of the optimisation step (BASELINE config 1: plumbing on CPU) and the N>1 path on gloo.

The product generator has no CPU path, so these tests plug a TEST-ONLY oracle-backed generator
(autograd through the PyTorch oracle) behind the same `synthesis` interface; what is under test is the
host side: HeadNeRF_* wiring, label flip, loss, optimiser ordering, checkpoint round trip and the
flattened all-reduce of the shared gradients."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp
from torch import nn

from hfa_gp_amd import headnerf
from hfa_gp_amd.config import tiny14
from hfa_gp_amd.generator import TriPlaneGenerator
from hfa_gp_amd.trainer import FlatGrads, Trainer, allreduce_shared_grads, epoch_batches, fit_frames, shard_range
from oracle import eg3d_oracle as O
from tests.util import look_at_label


class OracleGenerator(TriPlaneGenerator):
    """TEST INFRASTRUCTURE: a TriPlaneGenerator (same parameters / state_dict keys) whose synthesis runs
    the CPU oracle with autograd.  Instances are made by re-classing an existing generator."""

    @classmethod
    def adopt(cls, gen, seed=0):
        gen.__class__ = cls
        gen.seed = seed
        gen.labels_seen = []
        return gen

    def synthesis(self, ws, c=None, noise_mode="const", u_strat=None, u_imp=None, **_):
        assert noise_mode == "const"
        self.labels_seen.append(c.detach().clone())
        P = dict(self.named_parameters())
        P.update(dict(self.named_buffers()))
        g = torch.Generator().manual_seed(self.seed)
        b, r = ws.shape[0], self.cfg.neural_rendering_resolution ** 2
        # every sample of a batch sees the SAME uniforms, so a frame renders identically at any batch position
        # (or the caller's: the `u_strat` / `u_imp` test hook of HeadNeRF_*.get_image / Trainer.gen_update)
        us = torch.rand(1, r, self.cfg.depth_resolution, 1, generator=g).expand(b, -1, -1, -1).contiguous()
        ui = torch.rand(r, self.cfg.depth_resolution_importance, generator=g).repeat(b, 1)
        if u_strat is not None:
            us = u_strat.reshape(b, r, self.cfg.depth_resolution, 1)
        if u_imp is not None:
            ui = u_imp.reshape(b * r, self.cfg.depth_resolution_importance)
        return O.synthesis(P, self.cfg, ws, c, us, ui)


class Args:
    out_pose = False
    person_2 = False
    params_len = 76
    size = 32
    batch_size = 1
    lr = 3e-4
    latent_dim_style = 512
    latent_dim_shape = 8
    generator_preset = "tiny14"
    generator_seed = 0


def make_trainer(seed=0, world_size=1, rank=0):
    torch.manual_seed(seed)
    gen = headnerf.HeadNeRF_3DMM(Args(), Args.size, "cpu", 512, Args.latent_dim_shape)
    OracleGenerator.adopt(gen.generator)
    return Trainer(Args(), "cpu", rank=rank, world_size=world_size, mode="3dmm", gen=gen, lpips="none")


def frame(seed):
    g = torch.Generator().manual_seed(seed)
    real = (0.5 * torch.randn(1, 3, 32, 32, generator=g)).clamp(-1, 1)
    params = torch.randn(1, 76, generator=g)
    label = look_at_label(torch.tensor([1.5]), torch.tensor([1.6]), flipped=False)
    return real, label, params


def test_gen_update_config1_plumbing():
    tr = make_trainer()
    g0 = {k: v.clone() for k, v in tr.gen.generator.state_dict().items()}
    b0 = tr.gen.bases.detach().clone()
    real, label, params = frame(2)
    before = label.clone()
    l2_3dmm, l2, lp, img = tr.gen_update(real, label, params)          # four values: train_3dmm.py:128 unpacks four
    assert l2_3dmm.shape == (1,) and float(l2_3dmm) == 0.0             # trainer_3dmm.py:53
    assert torch.isfinite(l2) and float(lp) == 0.0 and img.shape == (1, 3, 32, 32)
    flipped = before.clone()
    flipped[:, headnerf.FLIP_COLUMNS] *= -1
    assert torch.equal(label, flipped)                               # in-place flip reaches the caller
    assert tr.gen.bases.grad.abs().sum() > 0 and tr.gen.delta.grad.abs().sum() > 0
    assert all(p.grad is not None and p.grad.abs().sum() > 0 for p in tr.gen.weights_3dmm.parameters())
    assert not torch.equal(tr.gen.bases.detach(), b0)                # Adam stepped the shared basis
    assert all(torch.equal(v, g0[k]) for k, v in tr.gen.generator.state_dict().items())   # generator frozen
    # after tune_generator() the SAME optimiser starts moving the generator (trainer_rgb.py:58-60,69-71)
    tr.tune_generator()
    real, label, params = frame(3)
    tr.gen_update(real, label, params)
    moved = [k for k, v in tr.gen.generator.state_dict().items() if not torch.equal(v, g0[k])]
    assert any(k.endswith("conv1.weight") for k in moved)


def test_loss_decreases_on_one_frame():
    tr = make_trainer(seed=1)
    tr.optimizer = torch.optim.Adam([p for p in tr.gen.parameters() if p.requires_grad], lr=1e-2)
    real, label0, params = frame(5)
    losses = []
    for _ in range(6):
        losses.append(float(tr.gen_update(real, label0.clone(), params)[1]))
    assert losses[-1] < losses[0]


def test_checkpoint_round_trip(tmp_path):
    tr = make_trainer(seed=2)
    real, label, params = frame(7)
    tr.gen_update(real, label, params)
    path = tr.save(12, str(tmp_path))
    assert os.path.basename(path) == "000012.pt"
    sd = torch.load(path, weights_only=False)
    assert set(sd) == {"gen", "w_optim", "args"}                      # trainer_3dmm.py:113-121 (rgb mode: "g_optim")
    assert tr.w_optim is tr.optimizer and not hasattr(tr, "g_optim")
    assert "bases" in sd["gen"] and "delta" in sd["gen"] and "weights_3dmm.fc.0.weight" in sd["gen"]
    assert any(k.startswith("generator.") for k in sd["gen"])
    tr2 = make_trainer(seed=99)
    assert tr2.resume(path) == 12
    assert torch.equal(tr2.gen.bases.detach(), tr.gen.bases.detach())
    a = tr.sample(None, frame(8)[1], frame(8)[2])
    b = tr2.sample(None, frame(8)[1], frame(8)[2])
    assert torch.allclose(a, b, atol=1e-6)


def test_sample_bases_alternates_label_flip():
    tr = make_trainer(seed=3)
    imgs = tr.sample_bases()
    assert len(imgs) == Args.latent_dim_shape and imgs[0].shape == (1, 3, 64, 64)
    seen = tr.gen.generator.labels_seen
    assert torch.equal(seen[0], seen[2]) and not torch.equal(seen[0], seen[1])      # reference quirk 2
    assert torch.equal(seen[0][:, headnerf.FLIP_COLUMNS], -seen[1][:, headnerf.FLIP_COLUMNS])


def test_reference_format_3dmm_checkpoint_round_trip(tmp_path):
    """A checkpoint dict in the layout trainer_3dmm.py:113-121 writes ({"gen", "w_optim", "args"}, optimiser state of an
    Adam over ALL of gen.parameters()) resumes; and a "g_optim"-keyed one (trainer_rgb.py:143-151) is accepted too."""
    tr = make_trainer(seed=5)
    ref_optim = torch.optim.Adam(tr.gen.parameters(), lr=3e-4)        # what the reference pickles: fresh state or not
    torch.save({"gen": tr.gen.state_dict(), "w_optim": ref_optim.state_dict(), "args": Args()}, tmp_path / "000007.pt")
    tr2 = make_trainer(seed=6)
    assert tr2.resume(str(tmp_path / "000007.pt")) == 7
    assert torch.equal(tr2.gen.bases.detach(), tr.gen.bases.detach())
    torch.save({"gen": tr.gen.state_dict(), "g_optim": ref_optim.state_dict(), "args": Args()}, tmp_path / "000008.pt")
    assert make_trainer(seed=7).resume(str(tmp_path / "000008.pt")) == 8
    # and what this trainer saves has exactly the reference's keys and a param_groups entry over all parameters
    sd = torch.load(tr.save(9, str(tmp_path)), weights_only=False)
    assert len(sd["w_optim"]["param_groups"][0]["params"]) == len(list(tr.gen.parameters()))


def test_sample_bases_scale_follows_the_mode():
    """alpha = 5 e_i in trainer_3dmm.py:90, 10 e_i in trainer_rgb.py:120."""
    tr = make_trainer(seed=3)
    seen = []
    inner = tr.gen.get_latent
    tr.gen.get_latent = lambda w, p2=False: (seen.append(w.clone()), inner(w, p2))[1]
    tr.sample_bases()
    assert float(seen[0].max()) == 5.0 and float(seen[3][0, 3]) == 5.0
    tr.mode = "rgb"
    seen.clear()
    tr.sample_bases()
    assert float(seen[0].max()) == 10.0


def test_missing_lpips_warns_and_none_string_is_silent():
    import warnings
    torch.manual_seed(0)
    gen = headnerf.HeadNeRF_3DMM(Args(), Args.size, "cpu", 512, Args.latent_dim_shape)
    with pytest.warns(UserWarning, match="LPIPS"):
        Trainer(Args(), "cpu", mode="3dmm", gen=gen)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        Trainer(Args(), "cpu", mode="3dmm", gen=gen, lpips="none")


def test_flat_grads_alias_and_rebuild():
    """The .grad tensors ARE slices of one flat buffer: autograd accumulates into it, zero() clears it, and the buffer
    is rebuilt when the set of trainable parameters changes (tune_generator)."""
    tr = make_trainer(seed=4)
    flat = tr.flat_grads()
    shared = tr.shared_parameters()
    assert flat.numel == sum(p.numel() for p in shared) and flat.owns(shared)
    real, label, params = frame(2)
    tr.gen_update(real, label, params)
    assert tr.flat_grads() is flat                                        # still aliased after a step
    for p, off in zip(shared, flat.offsets):
        assert p.grad.data_ptr() == flat.flat.data_ptr() + 4 * off
        assert torch.equal(p.grad.reshape(-1), flat.flat[off: off + p.numel()])
    assert flat.flat.abs().sum() > 0
    flat.zero()
    assert tr.gen.bases.grad.abs().sum() == 0
    tr.tune_generator()
    flat2 = tr.flat_grads()
    assert flat2 is not flat and flat2.numel > flat.numel
    # round 6: every slice starts on a 16-byte boundary — the generator's 0-d noise strengths sit between conv weights —
    # the padding stays zero and is part of the buckets
    assert any(p.dim() == 0 for p in flat2.params) and flat2.size > flat2.numel
    assert all(off % 4 == 0 and p.grad.data_ptr() % 16 == flat2.flat.data_ptr() % 16 for p, off in zip(flat2.params, flat2.offsets))
    assert flat2.buckets[-1][1] == flat2.size
    # buckets tile the buffer
    fb = FlatGrads(shared, bucket_bytes=1 << 20)
    assert fb.buckets[0][0] == 0 and fb.buckets[-1][1] == fb.size
    assert all(a[1] == b[0] for a, b in zip(fb.buckets[:-1], fb.buckets[1:])) and len(fb.buckets) > 1


def test_shard_range_and_epoch_batches():
    """2000 frames over 8 ranks -> rank r owns [250 r, 250 (r+1)); ragged tails."""
    assert [shard_range(2000, r, 8) for r in (0, 1, 7)] == [(0, 250), (250, 500), (1750, 2000)]
    got = [shard_range(11, r, 4) for r in range(4)]
    assert got == [(0, 3), (3, 6), (6, 9), (9, 11)]
    assert shard_range(3, 3, 4) == (3, 3)                                  # more ranks than frames: empty shard
    for n, world, batch in [(11, 4, 2), (5, 2, 4), (3, 4, 1), (8, 2, 2), (0, 2, 2)]:
        per_rank = [list(epoch_batches(n, r, world, batch)) for r in range(world)]
        steps = {len(x) for x in per_rank}
        assert len(steps) == 1                                             # every rank joins every collective
        seen = sorted(int(i) for x in per_rank for idx, _ in x for i in idx)
        assert seen == list(range(n))                                      # each frame exactly once
        for s in range(steps.pop()):
            counts = [len(per_rank[r][s][0]) for r in range(world)]
            # mean over ranks of (weight_r * mean loss_r) == mean over the frames of the step
            for r in range(world):
                assert abs(per_rank[r][s][1] - counts[r] * world / sum(counts)) < 1e-12


def _fit_worker(rank, world, port, out, nframes=3):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        tr = make_trainer(seed=10 + rank, world_size=world, rank=rank)
        tr.optimizer = torch.optim.SGD(tr.gen.parameters(), lr=0.05)
        start = tr.gen.bases.detach().clone()
        reals, labels, params = _frame_set(nframes)
        losses = fit_frames(tr, reals, labels, params, epochs=1, batch=1)
        out[rank] = {"bases": tr.gen.bases.detach().clone(), "n": len(losses), "start": start,
                     "fc0": tr.gen.weights_3dmm.fc[0].weight.detach().clone()}
    finally:
        dist.destroy_process_group()


def _frame_set(n):
    fr = [frame(60 + i) for i in range(n)]
    return torch.cat([f[0] for f in fr]), torch.cat([f[1] for f in fr]), torch.cat([f[2] for f in fr])


def test_fit_frames_two_ranks_ragged_equals_single_rank_global_batch():
    """3 frames over 2 ranks, batch 1 per rank (shards [0,2) and [2,3)): step 0 trains on frames {0, 2}, step 1 on frame 1
    alone (rank 1 joins with an empty batch).  With SGD the result must equal a single process that takes the same two
    global batches — the weighted all-reduce mean is the gradient of the mean over the frames of the step."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_fit_worker, args=(world, port, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    assert r0["n"] == 2 and r1["n"] == 2
    assert torch.equal(r0["bases"], r1["bases"]) and torch.equal(r0["fc0"], r1["fc0"])
    tr = make_trainer(seed=10)
    tr.optimizer = torch.optim.SGD(tr.gen.parameters(), lr=0.05)
    start = tr.gen.bases.detach().clone()
    assert torch.equal(start, r0["start"])
    reals, labels, params = _frame_set(3)
    for idx in ([0, 2], [1]):
        tr.gen_update(reals[idx], labels[idx].clone(), params[idx])
    # (the test-only oracle generator gives every sample the same renderer uniforms, so the two runs are comparable:
    # compare the UPDATES, which are ~3e-4 on an O(1) basis)
    want, got = tr.gen.bases.detach() - start, r0["bases"] - start
    assert want.abs().max() > 1e-4
    assert (got - want).abs().max() <= 2e-3 * want.abs().max(), ((got - want).abs().max(), want.abs().max())


def test_fit_frames_four_ranks_ragged_equals_single_rank_global_batch():
    """The same with FOUR ranks (BASELINE config 4 shards frames over 8): 5 frames, batch 1 per rank — shards [0,2) [2,3) [3,4)
    [4,5): step 0 trains on frames {0, 2, 3, 4}, step 1 on frame 1 alone while three ranks join with empty batches."""
    from hfa_gp_amd.trainer import shard_range
    world, port = 4, _free_port()
    assert [shard_range(5, r, world) for r in range(world)] == [(0, 2), (2, 3), (3, 4), (4, 5)]
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_fit_worker, args=(world, port, out, 5), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    assert all(r["n"] == 2 for r in res)
    assert all(torch.equal(res[0]["bases"], r["bases"]) and torch.equal(res[0]["fc0"], r["fc0"]) for r in res[1:])
    tr = make_trainer(seed=10)
    tr.optimizer = torch.optim.SGD(tr.gen.parameters(), lr=0.05)
    start = tr.gen.bases.detach().clone()
    reals, labels, params = _frame_set(5)
    for idx in ([0, 2, 3, 4], [1]):
        tr.gen_update(reals[idx], labels[idx].clone(), params[idx])
    want, got = tr.gen.bases.detach() - start, res[0]["bases"] - start
    assert want.abs().max() > 1e-4
    assert (got - want).abs().max() <= 2e-3 * want.abs().max(), ((got - want).abs().max(), want.abs().max())


# ----------------------------------------------------------------------------- N > 1 (gloo, world_size 2)
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        # different initial seeds per rank: the constructor must broadcast rank 0's parameters
        tr = make_trainer(seed=10 + rank, world_size=world, rank=rank)
        start = tr.gen.bases.detach().clone()
        real, label, params = frame(20 + rank)              # each rank owns its own frame
        tr.gen_update(real, label, params)
        out[rank] = {"start": start, "bases": tr.gen.bases.detach().clone(),
                     "grad": tr.gen.bases.grad.detach().clone(),
                     "fc0": tr.gen.weights_3dmm.fc[0].weight.detach().clone()}
    finally:
        dist.destroy_process_group()


def test_two_ranks_allreduce_shared_grads():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
        assert torch.equal(r0["start"], r1["start"])                      # broadcast from rank 0
        assert torch.equal(r0["grad"], r1["grad"])                        # one flattened all-reduce
        assert torch.equal(r0["bases"], r1["bases"]) and torch.equal(r0["fc0"], r1["fc0"])
    # the synchronised gradient is the MEAN of the two per-frame gradients
    grads = []
    for rank in range(2):
        tr = make_trainer(seed=10, world_size=1)
        real, label, params = frame(20 + rank)
        tr.optimizer = torch.optim.SGD(tr.gen.parameters(), lr=0.0)
        tr.gen_update(real, label, params)
        grads.append(tr.gen.bases.grad.detach().clone())
    assert torch.allclose(r0["grad"], 0.5 * (grads[0] + grads[1]), atol=1e-7, rtol=1e-4)


def _flat_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = torch.nn.Parameter(torch.zeros(3, 4))
        b = torch.nn.Parameter(torch.zeros(5))
        c = torch.nn.Parameter(torch.zeros(2), requires_grad=False)
        a.grad = torch.full((3, 4), float(rank + 1))
        n = allreduce_shared_grads([a, b, c], world)        # b has no grad yet -> treated as zeros
        out[rank] = (n, a.grad.clone(), b.grad.clone())
    finally:
        dist.destroy_process_group()


def test_flat_allreduce_numerics():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_flat_worker, args=(world, port, out), nprocs=world, join=True)
        for rank in range(world):
            n, ga, gb = out[rank]
            assert n == 17 and torch.equal(ga, torch.full((3, 4), 1.5)) and torch.equal(gb, torch.zeros(5))


def test_lpips_alex_module_and_objective():
    """LPIPS(alex) stand-in: lpips-package key names, zero for identical images, used as the reference's second loss
    term (trainer_rgb.py:86-91: g_loss = l2 + mean(lpips))."""
    from hfa_gp_amd.lpips_alex import LPIPSAlex
    torch.manual_seed(0)
    lp = LPIPSAlex()
    keys = list(lp.state_dict())
    for k in ("scaling_layer.shift", "net.slice1.0.weight", "net.slice2.3.bias", "net.slice5.10.weight", "lin0.model.1.weight",
              "lin4.model.1.weight"):
        assert k in keys, k
    assert all(not p.requires_grad for p in lp.parameters())
    a = torch.rand(2, 3, 64, 64) * 2 - 1
    b = torch.rand(2, 3, 64, 64) * 2 - 1
    assert lp(a, a).abs().max().item() == 0.0
    d = lp(a, b)
    assert d.shape == (2, 1, 1, 1) and bool((d > 0).all())
    assert torch.allclose(d, lp(b, a), atol=1e-6)
    lp2 = LPIPSAlex(state_dict=lp.state_dict())
    assert torch.equal(lp2(a, b), d)
    # an in1 at twice the resolution: the 2 x 2 average pool of the reference (trainer_rgb.py:84) folded into the first conv
    big = (torch.rand(2, 3, 128, 128) * 2 - 1).requires_grad_(True)
    v_pool = lp(a, torch.nn.functional.adaptive_avg_pool2d(big, 64)).sum()
    v_fold = lp(a, big).sum()
    (g_pool,), (g_fold,) = torch.autograd.grad(v_pool, big), torch.autograd.grad(v_fold, big)
    assert abs(float(v_pool) - float(v_fold)) <= 1e-5 * abs(float(v_pool))
    assert (g_pool - g_fold).abs().max() <= 1e-4 * g_pool.abs().max()
    # as the second term of the step
    torch.manual_seed(4)
    gen = headnerf.HeadNeRF_3DMM(Args(), Args.size, "cpu", 512, Args.latent_dim_shape)
    OracleGenerator.adopt(gen.generator)
    tr = Trainer(Args(), "cpu", mode="3dmm", gen=gen, lpips=lp)
    real, label, params = frame(9)
    _, l2, lpv, _ = tr.gen_update(real, label, params)
    assert float(lpv) > 0 and torch.isfinite(l2) and tr.gen.bases.grad.abs().sum() > 0


# ----------------------------------------------------------------------------- audio-driven trainer
class AudioArgs(Args):
    params_len = 64          # train_audio.py:186
    dim_aud = 64             # :206
    win_size = 16            # :208
    nosmo_iters = 5          # :210 (300000 in the reference; small so both branches run)
    smo_size = 8             # :212


def make_audio_trainer(seed=0, n=12, i_train=10, world_size=1, rank=0):
    from hfa_gp_amd.trainer import AudioTrainer
    torch.manual_seed(seed)
    gen = headnerf.HeadNeRF_Audio(AudioArgs(), AudioArgs.size, "cpu", 512, AudioArgs.latent_dim_shape)
    OracleGenerator.adopt(gen.generator)
    auds = torch.randn(n, 16, 29, generator=torch.Generator().manual_seed(50)).numpy()      # BASELINE config 5 shape
    return AudioTrainer(auds, i_train, AudioArgs(), "cpu", rank=rank, world_size=world_size, gen=gen, lpips="none")


def test_audio_window_padding_matches_reference_semantics():
    """trainer_audio.py:66-83: window [i-4, i+4) clipped to [0, limit), zero rows put back on the clipped side."""
    from hfa_gp_amd.trainer import audio_window, audio_windows
    auds = torch.arange(12.0)[:, None, None].expand(12, 16, 29) + 1.0          # frame i holds the value i+1
    w = audio_window(auds, 1, 8, 10)
    assert [int(v) for v in w[:, 0, 0]] == [0, 0, 0, 1, 2, 3, 4, 5]
    w = audio_window(auds, 8, 8, 10)
    assert [int(v) for v in w[:, 0, 0]] == [5, 6, 7, 8, 9, 10, 0, 0]          # limit = i_train = 10 hides frames 10, 11
    w = audio_window(auds, 8, 8, 12)
    assert [int(v) for v in w[:, 0, 0]] == [5, 6, 7, 8, 9, 10, 11, 12]
    idx = torch.tensor([0, 1, 5, 8, 11])
    batched = audio_windows(auds, idx, 8, 12)
    for n, i in enumerate(idx.tolist()):
        assert torch.equal(batched[n], audio_window(auds, i, 8, 12))
    # the centre row of the window is the frame itself
    assert torch.equal(batched[:, 4, 0, 0], idx.float() + 1.0)


def test_audio_attention_batched_equals_per_window():
    torch.manual_seed(0)
    att = headnerf.AudioAttNet()
    x = torch.randn(5, 8, 64)
    want = torch.stack([att(x[i]) for i in range(5)])
    assert torch.allclose(att.forward_windows(x), want, atol=1e-6)


def test_audio_trainer_step_both_branches_and_checkpoint(tmp_path):
    tr = make_audio_trainer()
    real, label, _ = frame(4)
    att0 = [p.detach().clone() for p in tr.AudAttNet.parameters()]
    aud0 = [p.detach().clone() for p in tr.AudNet.parameters()]
    # before nosmo_iters: AudioNet on the single frame, the attention net is neither used nor stepped
    out = tr.gen_update(real, label.clone(), None, global_step=0, img_i=3)
    assert len(out) == 4 and float(out[0]) == 0.0 and torch.isfinite(out[1]) and out[3].shape == (1, 3, 32, 32)
    assert all(torch.equal(a, b.detach()) for a, b in zip(att0, tr.AudAttNet.parameters()))
    assert any(not torch.equal(a, b.detach()) for a, b in zip(aud0, tr.AudNet.parameters()))
    assert tr.gen.bases.grad.abs().sum() > 0
    # after: smoothing window + attention, all three optimisers step
    tr.gen_update(real, label.clone(), None, global_step=5, img_i=0)             # window clipped on the left
    assert any(not torch.equal(a, b.detach()) for a, b in zip(att0, tr.AudAttNet.parameters()))
    path = tr.save(7, str(tmp_path))
    sd = torch.load(path, weights_only=False)
    assert set(sd) == {"gen", "AudAttNet", "AudNet", "w_optim", "optimizer_Aud", "optimizer_AudAtt", "args"}
    tr2 = make_audio_trainer(seed=9)
    assert tr2.resume(path) == 7
    a = tr.sample(None, label.clone(), None, 5, 11)                              # clipped on the right (limit = len(auds))
    b = tr2.sample(None, label.clone(), None, 5, 11)
    assert torch.allclose(a, b, atol=1e-6)
    # the batched reenactment path renders the same frames as the per-frame loop
    idx = torch.tensor([0, 6, 11])
    lab = label.repeat(3, 1)
    with torch.no_grad():
        drv = tr.drive_frames(idx)
        for n, i in enumerate(idx.tolist()):
            assert torch.allclose(drv[n:n + 1], tr._drive(5, i, tr.auds.shape[0]), atol=1e-6)
    batched = tr.sample_frames(idx, lab.clone())
    assert batched.shape == (3, 3, 64, 64) and torch.isfinite(batched).all()
    # (the test-only oracle generator draws its sampling uniforms per call, so images are compared loosely)
    one = tr.sample(None, label.clone(), None, 5, 6)
    assert (batched[1:2] - one).abs().max() < 5e-2


def _audio_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        tr = make_audio_trainer(seed=30 + rank, world_size=world, rank=rank)
        real, label, _ = frame(40 + rank)
        tr.gen_update(real, label, None, global_step=5, img_i=2 + 5 * rank)       # contiguous shards: own frame each
        out[rank] = {"bases": tr.gen.bases.detach().clone(),
                     "aud": next(tr.AudNet.parameters()).detach().clone(),
                     "att_grad": next(tr.AudAttNet.parameters()).grad.detach().clone()}
    finally:
        dist.destroy_process_group()


def test_audio_trainer_two_ranks():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_audio_worker, args=(world, port, out), nprocs=world, join=True)
        for k in ("bases", "aud", "att_grad"):
            assert torch.equal(out[0][k], out[1][k]), k


# ----------------------------------------------------------------------------- bucket order is rank-independent
class RgbArgs(Args):
    person_2 = True
    same_bases = False
    init = False
    out_pose = True


def make_rgb_trainer(seed=0, world_size=1, rank=0, bucket_bytes=256 << 10):
    torch.manual_seed(seed)
    gen = headnerf.HeadNeRF_final(RgbArgs(), RgbArgs.size, "cpu", 512, RgbArgs.latent_dim_shape)
    OracleGenerator.adopt(gen.generator)
    tr = Trainer(RgbArgs(), "cpu", rank=rank, world_size=world_size, mode="rgb", gen=gen, lpips="none")
    tr.bucket_bytes = bucket_bytes
    return tr


def _rgb_ragged_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        tr = make_rgb_trainer(seed=10 + rank, world_size=world, rank=rank)
        tr.optimizer = torch.optim.SGD(tr.gen.parameters(), lr=0.05)
        reals, labels, _ = _frame_set(3)
        orders = []
        fit_frames(tr, reals, labels, epochs=1, batch=1, on_step=lambda i, o: orders.append(list(tr._bucketer.last_order)))
        tr.tune_generator()                                   # more buckets, generator gradients through the sink
        fit_frames(tr, reals, labels, epochs=1, batch=1, on_step=lambda i, o: orders.append(list(tr._bucketer.last_order)))
        out[rank] = {"orders": orders, "bases": tr.gen.bases.detach().clone(),
                     "enc": tr.gen.encoder.fc[0].weight.detach().clone(),
                     "gw": tr.gen.generator.backbone.synthesis.b8.conv1.weight.detach().clone(),
                     "nb": len(tr._flat.buckets)}
    finally:
        dist.destroy_process_group()


def test_bucket_launch_order_is_rank_independent_with_empty_batches():
    """3 frames / 2 ranks / batch 1 in RGB mode with buckets small enough for many of them: at step 1 rank 1 has an EMPTY
    batch (no backward pass) while rank 0 finishes gradients in backward order.  Both must issue the SAME collective
    sequence (index order) — with readiness-order launching gloo aborts with a size mismatch here — and end
    with identical parameters, also once the generator is tuned."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_rgb_ragged_worker, args=(world, port, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    assert r0["nb"] > 3
    assert r0["orders"] == r1["orders"]
    for order in r0["orders"]:
        assert order == list(range(len(order)))
    assert len({len(o) for o in r0["orders"][2:]}) == 1 and len(r0["orders"][2]) == r0["nb"]
    for k in ("bases", "enc", "gw"):
        assert torch.equal(r0[k], r1[k]), k


def test_gradient_for_a_parameter_declared_absent_is_refused():
    """a non-generator parameter listed by `absent_parameters` that receives a gradient after all would be summed
    into a bucket that is already in flight (ranks diverge silently) and hidden from Adam by `step_skipping`: the bucketer
    raises instead.  Forced here by declaring the ACTIVE identity's basis absent."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        tr = make_rgb_trainer(seed=3, world_size=1)
        tr.force_collective = True
        real, label, _ = frame(7)
        tr.gen_update(real, label.clone(), False)                            # a legal step first
        honest = tr.absent_parameters
        tr.absent_parameters = lambda person_2=False: honest(person_2) + [tr.gen.bases]
        with pytest.raises(RuntimeError, match="declared ABSENT"):
            tr.gen_update(real, label.clone(), False)
        tr.absent_parameters = honest
        tr.gen_update(real, label.clone(), False)                            # and the trainer is usable afterwards
        assert not tr._bucketer.active
    finally:
        dist.destroy_process_group()


def test_bucketer_overlap_state_and_absent_parameters():
    """One-rank gloo group: (i) buckets leave in index order while the backward pass is still running (the flat buffer is in
    readiness order, so bucket 0 is NOT last); (ii) an exception between forward and backward leaves no stale counters;
    (iii) outside gen_update the generator has no gradient sink; (iv) parameters without a gradient this step (the other
    identity's basis, the pose head) take no Adam step — the reference's zero_grad(set_to_none) semantics."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        tr = make_rgb_trainer(seed=3, world_size=1)
        tr.force_collective = True
        tr.tune_generator()
        real, label, _ = frame(7)
        orig = tr.gen.generator.synthesis

        b2_before = tr.gen.bases_2.detach().clone()
        pose_before = [p.detach().clone() for p in tr.gen.encoder.pose.parameters()]
        tr.gen_update(real, label.clone(), False)
        bk = tr._bucketer
        assert bk.last_order == list(range(len(tr._flat.buckets))) and not bk.active
        assert bk.launched_early == len(tr._flat.buckets)                   # (i) all of them left during the backward pass
        assert getattr(tr.gen.generator, "_grad_sink", None) is None
        assert torch.equal(tr.gen.bases_2, b2_before)                       # absent: no Adam step at all
        assert all(torch.equal(a, b) for a, b in zip(pose_before, tr.gen.encoder.pose.parameters()))
        st = tr.optimizer.state
        assert tr.gen.bases in st and tr.gen.bases_2 not in st
        assert tr._flat.owns(tr.shared_parameters())                        # the hidden .grad slices are back
        # the other identity: now bases / delta are absent and bases_2 / delta_2 move
        b1_before = tr.gen.bases.detach().clone()
        tr.gen_update(real, label.clone(), True)
        assert torch.equal(tr.gen.bases, b1_before) and not torch.equal(tr.gen.bases_2, b2_before)

        # (ii) a step that dies after the forward pass
        def boom(*a, **k):
            raise RuntimeError("boom")
        tr.gen.generator.synthesis = boom
        with pytest.raises(RuntimeError, match="boom"):
            tr.gen_update(real, label.clone(), False)
        tr.gen.generator.synthesis = orig
        assert not bk.active and getattr(tr.gen.generator, "_grad_sink", None) is None
        tr.gen_update(real, label.clone(), False)
        assert tr._bucketer.last_order == list(range(len(tr._flat.buckets)))
        # (iii) plain autograd outside a step reaches the generator parameters
        w = tr.gen.get_weights(real)[0]
        img = tr.gen.get_image(tr.gen.get_latent(w, False), label.clone())
        gw = torch.autograd.grad(img.square().mean(), tr.gen.generator.backbone.synthesis.b8.conv1.weight)[0]
        assert gw.abs().max() > 0
    finally:
        dist.destroy_process_group()


def test_trainable_basis_is_never_served_from_the_q_cache():
    """Round-5 find (GPU): torch's fused Adam updates parameters without advancing `_version`, the key of the Q cache — so for a
    TRAINABLE basis the cache would serve the first `sample()` call's Q for ever.  A basis that requires grad is re-factorised on
    every call, also under no_grad; a frozen one is cached.  (The version-less update is emulated with a `.data` write.)"""
    torch.manual_seed(5)
    gen = headnerf.HeadNeRF_3DMM(Args(), Args.size, "cpu", 512, Args.latent_dim_shape)
    OracleGenerator.adopt(gen.generator)
    with torch.no_grad():
        q0 = gen._orthonormal(gen.bases).clone()
        v0 = gen.bases._version
        gen.bases.data.add_(0.05 * torch.randn_like(gen.bases))           # what fused Adam does: new values, same _version
        assert gen.bases._version == v0
        q1 = gen._orthonormal(gen.bases)
    assert (q1 - q0).abs().max().item() > 1e-4                           # the new basis, not the cached factor
    want = torch.linalg.qr((gen.bases.detach() + 1e-8).T, mode="reduced")[0]
    assert torch.allclose(q1, want, atol=1e-5)
    gen.bases.requires_grad_(False)                                      # frozen (reenactment): cached
    with torch.no_grad():
        a = gen._orthonormal(gen.bases)
        b = gen._orthonormal(gen.bases)
    assert a is b


def test_q_cache_of_a_basis_trained_and_then_frozen_on_the_same_object():
    """The `_version` class of bug, latent-basis cache: a frozen basis is cached, then made trainable, moved by a
    version-less update (what torch's fused Adam does), and frozen again — the cached Q of the first frozen phase has the same key
    (id, `_version`, `data_ptr`) and would be served.  Any use while trainable empties the cache."""
    torch.manual_seed(6)
    gen = headnerf.HeadNeRF_3DMM(Args(), Args.size, "cpu", 512, Args.latent_dim_shape)
    OracleGenerator.adopt(gen.generator)
    gen.bases.requires_grad_(False)
    with torch.no_grad():
        q_frozen = gen._orthonormal(gen.bases)
    gen.bases.requires_grad_(True)
    gen._orthonormal(gen.bases)                                          # a training step's use
    v0 = gen.bases._version
    gen.bases.data.add_(0.05 * torch.randn_like(gen.bases))
    assert gen.bases._version == v0
    gen.bases.requires_grad_(False)
    with torch.no_grad():
        q_after = gen._orthonormal(gen.bases)
    want = torch.linalg.qr((gen.bases.detach() + 1e-8).T, mode="reduced")[0]
    assert torch.allclose(q_after, want, atol=1e-5) and (q_after - q_frozen).abs().max().item() > 1e-4


def test_multi_tensor_adam_decides_its_fallback_before_the_closure_runs():
    """with an option the one-launch kernel does not implement (weight decay) `MultiTensorAdam.step(closure)` used to
    evaluate the closure, then call torch's step WITHOUT it and return None.  Now torch's own step gets the closure: one evaluation,
    its loss returned.  (CPU tensors: the fallback path needs no GPU.)"""
    from hfa_gp_amd.trainer import MultiTensorAdam
    p = torch.nn.Parameter(torch.ones(4))
    opt = MultiTensorAdam([p], lr=0.1, weight_decay=0.01)
    calls = []

    def closure():
        opt.zero_grad()
        loss = (p * p).sum()
        loss.backward()
        calls.append(1)
        return loss
    out = opt.step(closure)
    assert len(calls) == 1 and out is not None and abs(float(out) - 4.0) < 1e-6
    assert not torch.equal(p.detach(), torch.ones(4))


def test_equal_linear_with_the_scale_folded_into_the_gemm_matches_the_plain_expression():
    """encoder3d._EqualLinearFn (the 3DMM driver's layers: `F.linear(x, W * scale, b * lr_mul)` without the four scalar-multiply
    kernels per layer and step): values and all three gradients in fp64, lr_mul = 1 and != 1; 3-D inputs keep the plain path."""
    import torch.nn.functional as F
    from hfa_gp_amd.encoder3d import EqualLinear
    torch.manual_seed(0)
    for lr in (1, 0.01):
        lin = EqualLinear(50, 70, lr_mul=lr).double()
        lin.bias.data.normal_()
        x = torch.randn(3, 50, dtype=torch.float64, requires_grad=True)
        y = lin(x)
        gy = torch.randn_like(y)
        got = torch.autograd.grad(y, (x, lin.weight, lin.bias), gy)
        ref_y = F.linear(x, lin.weight * lin.scale, bias=lin.bias * lin.lr_mul)
        ref = torch.autograd.grad(ref_y, (x, lin.weight, lin.bias), gy)
        assert (y - ref_y).abs().max().item() < 1e-12
        for a, b in zip(got, ref):
            assert (a - b).abs().max().item() < 1e-12
        x3 = torch.randn(2, 3, 50, dtype=torch.float64)
        assert (lin(x3) - F.linear(x3, lin.weight * lin.scale, bias=lin.bias * lin.lr_mul)).abs().max().item() < 1e-12
