"""Batched reenactment harness — counterpart of the per-frame loop in
/root/reference/code/run_recon_video_rgb.py:216-236 (driver → get_latent → get_image → save_image)
and of its `layout_grid` uint8 quantisation (:28-42).  PNG / mp4 encoding is out of scope
(SURVEY.md §2.1 row 7); frames come back as uint8 tensors, quantised on the GPU and copied to pinned
host memory asynchronously so the copy of batch i overlaps the render of batch i+1."""
from __future__ import annotations

from typing import Iterable, Iterator, Optional, Tuple

import torch


def to_uint8(img: torch.Tensor) -> torch.Tensor:
    """[-1, 1] float image → uint8 with the reference's rounding: (img * 127.5 + 128).clamp(0, 255)."""
    return (img * 127.5 + 128).clamp(0, 255).to(torch.uint8)


def layout_grid(img: torch.Tensor, grid_w: Optional[int] = None, grid_h: int = 1, float_to_uint8: bool = True,
                chw_to_hwc: bool = True, to_numpy: bool = True):
    """Tile a batch [B,C,H,W] into one grid_h x grid_w image."""
    b, c, h, w = img.shape
    if grid_w is None:
        grid_w = b // grid_h
    assert b == grid_w * grid_h
    if float_to_uint8:
        img = to_uint8(img)
    img = img.reshape(grid_h, grid_w, c, h, w).permute(2, 0, 3, 1, 4).reshape(c, grid_h * h, grid_w * w)
    if chw_to_hwc:
        img = img.permute(1, 2, 0)
    return img.cpu().numpy() if to_numpy else img


@torch.no_grad()
def render_frames(gen, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], person_2: bool = False,
                  to_host: bool = True) -> Iterator[torch.Tensor]:
    """`gen` is a HeadNeRF_* module; `batches` yields (driver_input [B,...], label [B,25]) on gen's device.
    Yields uint8 frames [B,3,H,W].  The label flip side effect of `get_image` is preserved."""
    gen.eval()
    copy_stream = torch.cuda.Stream() if to_host and torch.cuda.is_available() else None
    pending = None
    for driver_in, label in batches:
        weights = gen.get_weights(driver_in)
        if isinstance(weights, tuple):
            weights = weights[0]
        frames = to_uint8(gen.get_image(gen.get_latent(weights, person_2), label))
        if copy_stream is None:
            yield frames
            continue
        host = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True)
        copy_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(copy_stream):
            host.copy_(frames, non_blocking=True)
            frames.record_stream(copy_stream)
            done = torch.cuda.Event()
            done.record(copy_stream)
        if pending is not None:
            pending[1].synchronize()
            yield pending[0]
        pending = (host, done)
    if pending is not None:
        pending[1].synchronize()
        yield pending[0]
