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

from concurrent.futures import ThreadPoolExecutor

import torch

from . import parallel

_HOST_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix="m324-stage")     # pageable frames -> pinned bounce buffers

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


_MERGE_INDEX: Dict[tuple, tuple] = {}              # (device, n_windows, C, out_map) -> (gather index on the device, frames replaced by ref_pcd)


def merge_windows(outs: torch.Tensor, out_map: List[Slot], ref_pcd: torch.Tensor) -> torch.Tensor:
    """outs [n_windows, C, N, 3] -> trajectories [1, len(out_map), N, 3]: one gather over the (window, slot) pairs of the plan
    (the index lives on the device, cached per plan: no host-to-device copy between the last forward and the result), then the
    frames the reference overwrites with ref_pcd."""
    nW, C = outs.shape[0], outs.shape[1]
    key = (str(outs.device), nW, C, tuple(out_map))
    hit = _MERGE_INDEX.get(key)
    if hit is None:
        if len(_MERGE_INDEX) >= 16:
            _MERGE_INDEX.clear()
        flat = [0 if s is None else s[0] * C + s[1] for s in out_map]
        hit = _MERGE_INDEX[key] = (torch.tensor(flat, dtype=torch.long).to(outs.device), [t for t, s in enumerate(out_map) if s is None])
    idx, from_ref = hit
    merged = outs.reshape(nW * C, *outs.shape[2:]).index_select(0, idx)
    for t in from_ref:
        merged[t] = ref_pcd.reshape(-1, 3).to(outs.dtype)
    return merged.unsqueeze(0)


def _cfg_get(cfg, key, default=None):
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


class _WindowFeeder:
    """Frames of the windows on their way to the GPU: window w + 1 uploads on a copy stream while window w computes.

    Source: the caller's video [T,H,W,3] -- fp32 in [0,1] or uint8 in 0..255 (a quarter of the bytes; m324_patchify_u8 converts
    every tap as v / 255, bit-identical to `video.float() / 255`), on the host (pinned: DMA straight from the caller's pages;
    pageable: through two pinned bounce buffers) or already on the device.  Two device staging buffers; an event per buffer says
    "uploaded" (the compute stream waits on it) and "consumed" (the next upload into that buffer waits on it).  The anchor frame
    0 opens every window of the reference's plan (scripts/inference_with_video_mesh.py:187-194): it is uploaded once and kept."""

    def __init__(self, dtype: torch.dtype, frame: tuple, device, max_frames: int):
        self.dev, self.max_frames = torch.device(device), max_frames
        self.stage = [torch.empty((max_frames,) + tuple(frame), dtype=dtype, device=self.dev) for _ in range(2)]
        self.bounce = None
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        # the staging buffers were allocated on the caller's stream and are written on the copy stream: whatever that stream still
        # has queued (a previous owner of the memory) comes first
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.dev))
        self.uploaded = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [None, None]
        self.host_done = [None, None]                          # bounce buffer k may be refilled once its DMA has left it
        self.video, self.on_host, self.anchor = None, False, None
        self.n = 0

    def begin(self, video: torch.Tensor) -> "_WindowFeeder":
        """A new video through the same staging buffers.  The events of the previous video stay in force: a caller that runs
        videos back to back without synchronising (the next video's first upload may be issued while the previous video's last
        window is still reading its frames) is ordered by them."""
        self.video, self.on_host, self.anchor = video, video.device.type == "cpu", None
        if self.on_host and not video.is_pinned() and self.bounce is None:
            self.bounce = [torch.empty(self.stage[0].shape, dtype=video.dtype).pin_memory() for _ in range(2)]
        self.pageable = self.on_host and not video.is_pinned()
        return self

    def _copy(self, slot: int, dst_off: int, lo: int, hi: int) -> None:
        dst = self.stage[slot][dst_off:dst_off + hi - lo]
        if self.pageable:
            # host memcpy into pinned pages, four ranges at a time (copy_ releases the GIL; one thread moves 3-10 GB/s, a window of
            # fp32 frames is 100 MB), then one DMA
            b = self.bounce[slot][dst_off:dst_off + hi - lo]
            n = hi - lo
            cuts = [lo + n * k // 4 for k in range(5)]
            jobs = [_HOST_POOL.submit(b[a - lo:e - lo].copy_, self.video[a:e]) for a, e in zip(cuts[:-1], cuts[1:]) if e > a]
            for j in jobs:
                j.result()
            dst.copy_(b, non_blocking=True)
        else:
            dst.copy_(self.video[lo:hi], non_blocking=True)

    def upload(self, frames) -> int:
        """Starts the upload of the given video frame indices (the anchor 0 may lead; the rest is one contiguous run) into the
        next staging buffer; returns the slot."""
        slot = self.n & 1
        self.n += 1
        st = self.copy_stream
        if self.pageable and self.host_done[slot] is not None:
            self.host_done[slot].synchronize()
        with torch.cuda.stream(st):
            if self.consumed[slot] is not None:
                st.wait_event(self.consumed[slot])
            off = 0
            rest = list(frames)
            if len(rest) > 1 and rest[0] == 0 and rest[1] != 1:           # anchor + a later run
                if self.anchor is None:
                    self.anchor = self.video[0:1].to(self.dev, non_blocking=True) if self.on_host else self.video[0:1].clone()
                self.stage[slot][0:1].copy_(self.anchor, non_blocking=True)
                off, rest = 1, rest[1:]
            lo, hi = rest[0], rest[-1] + 1
            assert rest == list(range(lo, hi)), "a window is the anchor frame plus one contiguous run of frames"
            self._copy(slot, off, lo, hi)
            self.uploaded[slot].record(st)
            if self.pageable:
                self.host_done[slot] = torch.cuda.Event()
                self.host_done[slot].record(st)
        return slot

    def frames(self, slot: int, n: int) -> torch.Tensor:
        """The staged frames [n,H,W,3] of a slot, valid for work enqueued on the current stream from here on."""
        torch.cuda.current_stream(self.dev).wait_event(self.uploaded[slot])
        return self.stage[slot][:n]

    def release(self, slot: int) -> None:
        """Everything enqueued so far on the current stream has read the slot's frames."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self.consumed[slot] = ev


_FEEDERS: Dict[tuple, _WindowFeeder] = {}          # staging buffers (2 x one window of frames) kept per device and frame format


def _feeder_for(video: torch.Tensor, device, max_frames: int) -> _WindowFeeder:
    if video.dtype != torch.uint8 and video.dtype != torch.float32:
        video = video.float()                          # the reference's `.float()` (fp16 / fp64 frames)
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), video.dtype, tuple(video.shape[1:]), max_frames)
    f = _FEEDERS.get(key)
    if f is None:
        while len(_FEEDERS) >= 2:                      # a caller that keeps changing formats: drop the oldest
            del _FEEDERS[next(iter(_FEEDERS))]
        f = _FEEDERS[key] = _WindowFeeder(video.dtype, tuple(video.shape[1:]), dev, max_frames)
    return f.begin(video)


def _native(model) -> bool:
    from .Pcd_motion import Motion_Latent_Model
    return isinstance(model, Motion_Latent_Model)


def run_model_inference(model, input_data: Dict[str, torch.Tensor], video_tensor: torch.Tensor, config, device,
                        group=None, pipelined: Optional[bool] = None, reuse: Optional[bool] = None) -> Optional[torch.Tensor]:
    """Same contract as the reference driver: video_tensor [T,H,W,3] in [0,1] (any T) -> trajectories [1,T,N,3]
    fp32 on `device`.  With an initialised process group the windows are sharded over its ranks and the result is
    available on every rank.

    On a HIP device the windows are PIPELINED (pipelined=None: whenever the device is one; False: the plain loop, one
    synchronous upload per window, kept for A/B and as the reference's literal form): window w + 1's frames travel on a copy
    stream under window w's forward (_WindowFeeder: pinned staging, the anchor frame uploaded once, uint8 videos accepted --
    a quarter of the host-to-device bytes).  reuse (None: with this package's model): the shape encoder's latent tokens and
    the anchor frame's image tokens are computed by the first window and handed to the others (Motion_Latent_Model._forward:
    `m324_mesh_tokens`, `m324_anchor_tokens`) -- they are the same in every window, bit for bit; the result equals the plain
    loop's exactly (tests/test_configs_gpu.py)."""
    tr = _cfg_get(config, "training")
    chunk = _cfg_get(tr, "frames", 12)
    use_amp = _cfg_get(tr, "use_amp", False)
    total_T = video_tensor.shape[0]
    windows, out_map = plan_windows(total_T, chunk)
    ref_pcd = input_data["ref_pcd"]
    N = ref_pcd.shape[1]
    dev_type = torch.device(device).type
    if pipelined is None:
        pipelined = dev_type == "cuda"
    if pipelined and dev_type != "cuda":
        raise ValueError("run_model_inference(pipelined=True) needs a HIP device")
    if video_tensor.dtype == torch.uint8 and not _native(model):
        raise ValueError("uint8 frames are converted by this package's model only (m324_patchify_u8); pass `video.float() / 255`")
    if reuse is None:
        reuse = pipelined and _native(model) and not getattr(model, "training", False) and ref_pcd.shape[0] == 1
    rank, world = parallel.world_info(group)
    mine = list(parallel.partition(len(windows), world, rank))

    def check(out) -> torch.Tensor:
        if not (isinstance(out, dict) and "pcd_moved" in out):
            raise RuntimeError("model returned no pcd_moved")
        return out["pcd_moved"].float()[0]

    def call(sample):
        with torch.no_grad(), torch.autocast(enabled=bool(use_amp), device_type=dev_type, dtype=torch.bfloat16):
            return model(sample)

    if not pipelined:
        def forward_window(w: int) -> torch.Tensor:
            idx = torch.as_tensor(windows[w], device=video_tensor.device)
            sample = dict(input_data)
            frames = video_tensor.index_select(0, idx)[None]
            sample["rgb_video"] = (frames if frames.dtype == torch.uint8 else frames.float()).to(device)
            return check(call(sample))
        outs = [forward_window(w) for w in mine]
    else:
        feeder = _feeder_for(video_tensor, device, len(windows[0]))
        kept = None                                    # (mesh tokens, anchor tokens) of this video, from the first window that ran
        outs = []

        def frames_of(w: int):
            # a later window with the anchor's tokens at hand uploads only the frames behind the anchor
            return windows[w][1:] if (kept is not None and w > 0 and len(windows) > 1) else windows[w]
        slot = feeder.upload(frames_of(mine[0])) if mine else None
        for i, w in enumerate(mine):
            fr = frames_of(w)
            sample = dict(input_data)
            sample["rgb_video"] = feeder.frames(slot, len(fr))[None]
            if kept is not None and len(fr) < len(windows[w]):
                sample["m324_mesh_tokens"], sample["m324_anchor_tokens"] = kept
            elif kept is not None:
                sample["m324_mesh_tokens"] = kept[0]
            elif reuse and len(mine) > 1:
                sample["m324_keep_reuse"] = True
            out = call(sample)
            feeder.release(slot)
            if kept is None and reuse and len(mine) > 1:
                kept = (out["reuse"]["mesh_tokens"], out["reuse"]["anchor_tokens"]) if windows[w][0] == 0 else None
            if i + 1 < len(mine):                      # the next window's frames travel under this window's forward
                slot = feeder.upload(frames_of(mine[i + 1]))
            outs.append(check(out))
    local = torch.stack(outs, dim=0) if outs else torch.zeros((0, len(windows[0]), N, 3), dtype=torch.float32, device=device)
    return merge_windows(parallel.all_gather_items(local, len(windows), group), out_map, ref_pcd.to(device))
