"""Long-video driver: the reference's sliding-window inference, with the windows spread over GPUs.

Restates `run_model_inference` of the reference's caller (scripts/inference_with_video_mesh.py:132-256; twin in
scripts/inference_with_video_only.py:380-506) as an explicit index plan, so that
  * the plan can be checked against golden index maps produced by the reference function itself
    (tests/golden/chunks.json, tests/golden/make_chunk_golden.py), and
  * the independent windows can be handed to different ranks (motion324_amd.parallel) -- the windows share
    nothing but frame 0 of the video, so this is the natural multi-GPU work list for one long clip.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import parallel

Slot = Optional[Tuple[int, int]]      # (window index, frame slot inside that window) or None = ref_pcd


def window_starts(total_T: int, chunk: int) -> List[int]:
    """reference :177-180 -- stride chunk-1, plus a last window flush with the end of the clip."""
    slide = chunk - 1
    starts = list(range(0, total_T - chunk + 1, slide))
    if starts and starts[-1] + chunk < total_T:
        starts.append(total_T - chunk)
    return starts


def plan_windows(total_T: int, chunk: int) -> Tuple[List[List[int]], List[Slot]]:
    """Returns (windows, out_map).  windows[w] = the `chunk` video-frame indices fed to forward w (window 0 is
    frames 0..chunk-1, later windows are the anchor frame 0 followed by chunk-1 new frames, reference :187-194).
    out_map[t] says where output frame t comes from (reference merge rules :219-254); None = the frame is
    overwritten with ref_pcd (:224,240,248).  A clip that fits one window is a single forward whose frame 0 is
    kept as predicted (:157-174)."""
    if total_T <= chunk:
        return [list(range(total_T))], [(0, t) for t in range(total_T)]
    starts = window_starts(total_T, chunk)
    windows = [list(range(chunk)) if i == 0 else [0] + list(range(s + 1, s + chunk)) for i, s in enumerate(starts)]
    n = len(windows)
    out: List[Slot] = []
    if n == 1:
        return windows, [None] + [(0, t) for t in range(1, chunk)]
    for i in range(n):
        if i == 0 and n != 2:
            out += [None] + [(0, t) for t in range(1, chunk)]
        elif i < n - 2:
            out += [(i, t) for t in range(1, chunk)]
        elif i == n - 2:
            keep = max(starts[-1] - starts[-2], 0)
            if keep > 0 and n != 2:
                out += [(i, t) for t in range(1, 1 + keep)]
            elif keep > 0 and i == 0 and n == 2:
                out += [None] + [(0, t) for t in range(1, 1 + keep)]
        else:
            out += [(i, t) for t in range(1, chunk)]
    return windows, out


def merge_windows(outs: torch.Tensor, out_map: List[Slot], ref_pcd: torch.Tensor) -> torch.Tensor:
    """outs [n_windows, C, N, 3] -> trajectories [1, len(out_map), N, 3]."""
    frames = [ref_pcd.reshape(-1, 3).to(outs.dtype) if s is None else outs[s[0], s[1]] for s in out_map]
    return torch.stack(frames, dim=0).unsqueeze(0)


def _cfg_get(cfg, key, default=None):
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


def run_model_inference(model, input_data: Dict[str, torch.Tensor], video_tensor: torch.Tensor, config, device,
                        group=None) -> Optional[torch.Tensor]:
    """Same contract as the reference driver: video_tensor [T,H,W,3] in [0,1] (any T) -> trajectories [1,T,N,3]
    fp32 on `device`.  With an initialised process group the windows are sharded over its ranks and the result is
    available on every rank."""
    tr = _cfg_get(config, "training")
    chunk = _cfg_get(tr, "frames", 12)
    use_amp = _cfg_get(tr, "use_amp", False)
    total_T = video_tensor.shape[0]
    windows, out_map = plan_windows(total_T, chunk)
    ref_pcd = input_data["ref_pcd"]
    N = ref_pcd.shape[1]
    dev_type = torch.device(device).type

    def forward_window(w: int) -> torch.Tensor:
        idx = torch.as_tensor(windows[w], device=video_tensor.device)
        sample = dict(input_data)
        sample["rgb_video"] = video_tensor.index_select(0, idx)[None].float().to(device)
        with torch.no_grad(), torch.autocast(enabled=bool(use_amp), device_type=dev_type, dtype=torch.bfloat16):
            out = model(sample)
        if not (isinstance(out, dict) and "pcd_moved" in out):
            raise RuntimeError("model returned no pcd_moved")
        return out["pcd_moved"].float()[0]

    outs = parallel.map_items(forward_window, len(windows), (len(windows[0]), N, 3), device, group=group)
    return merge_windows(outs, out_map, ref_pcd.to(device))
