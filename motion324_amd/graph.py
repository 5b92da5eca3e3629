"""hipGraph replay of Motion_Latent_Model.forward.

One forward of the BASELINE clip enqueues ~420 kernels from Python (ctypes + torch allocator: about
7.6 ms of host time per clip against ~10-15 ms of GPU time).  The launch sequence is static for a
given input shape / precision, so it is captured once into a HIP graph (torch.cuda.CUDAGraph drives
hipStreamBeginCapture on the stream libm324 launches on) and replayed: one host call per clip.

    fast = GraphedForward(model)          # model.eval() on a HIP device
    out = fast(sample)                    # first call per (shapes, precision): warm-up + capture
    out.pcd_moved                         # static output buffer, overwritten by the next replay
    buf = fast.static_inputs(sample)      # zero-copy handover: fill buf[...] in place, then fast(buf)

Weights are read through the Prepared cache at capture time: a graph is keyed by the cache generation AND by a stamp
of every parameter's storage pointer and in-place version counter, so ``model.train()/eval()`` toggles, the native
optimizer step, ``load_state_dict`` / ``load_checkpoint`` on a live model and ``model.to(...)`` all force a re-capture.
Only an update through raw pointers that bypasses torch's version counters needs ``prepared.bump_generation()``.
"""
from __future__ import annotations

import gc
import weakref
from typing import Dict, Tuple

import torch

from .easydict import EasyDict as edict
from . import prepared
from .prepared import compute_dtype

_KEYS = ("ref_shape_pcd", "ref_shape_normals", "ref_shape_rgbs", "ref_pcd", "ref_normal", "ref_rgb", "rgb_video",
         "point_clouds")


class GraphedForward:
    def __init__(self, model: torch.nn.Module, warmup: int = 2, weak: bool = False):
        # weak: the model itself owns this object (Motion_Latent_Model's automatic graph replay) -- a strong reference back
        # would make model <-> graphs cyclic garbage, which Python's collector may free at any time, e.g. in the middle
        # of a LATER stream capture, where destroying a hipGraph aborts the process
        self._model = weakref.ref(model) if weak else (lambda: model)
        self.warmup = warmup
        self._graphs: Dict[Tuple, tuple] = {}

    @property
    def model(self) -> torch.nn.Module:
        return self._model()

    def reset(self) -> None:
        self._graphs.clear()

    def _key(self, sample) -> Tuple:
        return (prepared.generation(), prepared.weight_stamp(self.model), compute_dtype()) + \
            tuple((k, tuple(sample[k].shape)) for k in _KEYS if k in sample)

    def static_inputs(self, sample: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """The graph's own input buffers for this sample's shapes (captured on first use), holding a copy of `sample`.
        A producer that writes the next clip straight into these tensors and then calls ``fast(buffers)`` hands the clip
        over without the device-to-device copies ``fast(sample)`` makes (100.7 MB of frames for the BASELINE clip)."""
        self(sample)
        return self._graphs[self._key(sample)][1]

    def __call__(self, sample: Dict[str, torch.Tensor]):
        if self.model.training:
            raise RuntimeError("GraphedForward is an inference helper: call model.eval() first")
        key = self._key(sample)
        entry = self._graphs.get(key)
        if entry is None:
            for stale in [k for k in self._graphs if k[:2] != key[:2]]:   # graphs that point at dropped weight copies
                del self._graphs[stale]
            static_in = {k: sample[k].detach().to(torch.float32).contiguous().clone() for k in _KEYS if k in sample}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(self.warmup):          # populates the Prepared cache and the allocator pools
                    self.model(static_in)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            gc_was_on = gc.isenabled()
            gc.collect()
            gc.disable()                      # no collector run may free a stream / event / graph while the capture is open
            try:
                with torch.no_grad(), torch.cuda.graph(g):
                    static_out = self.model(static_in)
            finally:
                if gc_was_on:
                    gc.enable()
            entry = (g, static_in, static_out)
            self._graphs[key] = entry
        g, static_in, static_out = entry
        for k, buf in static_in.items():
            src = sample[k]
            if src.data_ptr() != buf.data_ptr():
                buf.copy_(src, non_blocking=True)
        g.replay()
        out = edict(input_data=sample, pcd_moved=static_out["pcd_moved"])
        if "loss_metrics" in static_out:
            out.loss_metrics = static_out["loss_metrics"]
        return out
