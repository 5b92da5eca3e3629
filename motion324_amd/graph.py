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
         "point_clouds", "m324_mesh_tokens", "m324_anchor_tokens")
_FLAGS = ("m324_keep_reuse",)          # non-tensor entries of a sample that change what the forward returns


def shape_key(sample) -> Tuple:
    """what a captured graph is specialised on besides weights and precision: every input's shape, whether the frames are
    bytes (m324_patchify_u8) or fp32, and the flags"""
    return tuple((k, tuple(sample[k].shape)) for k in _KEYS if k in sample) + \
        (("u8", sample["rgb_video"].dtype == torch.uint8),) + tuple((f, bool(sample.get(f, False))) for f in _FLAGS)


def _static_copy(k: str, t: torch.Tensor) -> torch.Tensor:
    if k == "rgb_video" and t.dtype == torch.uint8:
        return t.detach().contiguous().clone()
    return t.detach().to(torch.float32).contiguous().clone()


class _Segmenter:
    """A forward captured as a CHAIN of hipGraphs that share one memory pool, cut wherever the forward asks for an eager action
    between two of them (``cut(fn)``): the frame-parallel forward's RCCL exchanges.  Capturing a collective into a hipGraph
    segfaults on this stack (ROCm 7.0.2 / RCCL 2.26.6, tools/fp_graph_lab.py); an eager N > 1 forward leaves
    the GPU idle between its ~350 launches.  The chain replays graph, exchange, graph, ...: nine host calls plus eight
    collectives per clip instead of ~430 launches."""

    def __init__(self, capture_error_mode: str):
        self.pool = torch.cuda.graph_pool_handle()
        self.mode = capture_error_mode
        self.graphs, self.between, self.cur = [], [], None

    def begin(self) -> None:
        self.cur = torch.cuda.CUDAGraph()
        self.cur.capture_begin(pool=self.pool, capture_error_mode=self.mode)

    def cut(self, eager_fn) -> None:
        """Ends the open graph, runs eager_fn once (buffers only: nothing has executed yet), opens the next graph."""
        self.cur.capture_end()
        self.graphs.append(self.cur)
        self.cur = None
        eager_fn()
        self.between.append(eager_fn)
        self.begin()

    def end(self) -> None:
        if self.cur is not None:
            import warnings
            with warnings.catch_warnings():
                # the last link may hold no kernel at all (B = 1: the gathered output is already in its final layout)
                warnings.filterwarnings("ignore", message="The CUDA Graph is empty")
                self.cur.capture_end()
            self.graphs.append(self.cur)
            self.cur = None

    def replay(self) -> None:
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.between):
                self.between[i]()


_SEGMENTER = None


def active_segmenter():
    """The segmenter of the capture in progress on this process, if any (Pcd_motion asks before every collective)."""
    return _SEGMENTER


class GraphedForward:
    def __init__(self, model: torch.nn.Module, warmup: int = 2, weak: bool = False, max_graphs: int = 0,
                 capture_error_mode: str = "global", forward=None, segmented: bool = False):
        # forward: the callable captured instead of model(sample) -- e.g. model.forward_frame_parallel
        # segmented: capture a chain of graphs cut at the forward's collectives (_Segmenter) instead of one graph
        # max_graphs > 0: keep at most that many captured shape sets, dropping the least recently used (each graph owns a
        # private memory pool with a clip's activations and its static inputs).  capture_error_mode: torch.cuda.graph's
        # -- "thread_local" lets other threads (a DataLoader's pin-memory thread, another stream's allocation) keep
        # calling into HIP while this thread captures.  One override: a segmented capture with M324_KV_OVERLAP=1 (an
        # exchange in flight across a cut) turns "global" into "thread_local" (see __call__).
        # weak: the model itself owns this object (Motion_Latent_Model's automatic graph replay) -- a strong reference back
        # would make model <-> graphs cyclic garbage, which Python's collector may free at any time, e.g. in the middle
        # of a LATER stream capture, where destroying a hipGraph aborts the process
        self._model = weakref.ref(model) if weak else (lambda: model)
        self.warmup = warmup
        self.max_graphs = max_graphs
        self._forward = forward
        self.segmented = segmented
        self.capture_error_mode = capture_error_mode
        self._graphs: Dict[Tuple, tuple] = {}               # insertion order = recency (re-inserted on every use)

    @property
    def model(self) -> torch.nn.Module:
        return self._model()

    def reset(self) -> None:
        self._graphs.clear()

    def _key(self, sample) -> Tuple:
        return (prepared.generation(), prepared.weight_stamp(self.model), compute_dtype()) + shape_key(sample)

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
        run = self._forward if self._forward is not None else self.model
        if entry is None:
            for stale in [k for k in self._graphs if k[:2] != key[:2]]:   # graphs that point at dropped weight copies
                del self._graphs[stale]
            while self.max_graphs > 0 and len(self._graphs) >= self.max_graphs:
                del self._graphs[next(iter(self._graphs))]                 # least recently used
            # The static buffers and the captured forward live OUTSIDE inference mode, whatever mode the caller is in:
            # tensors created under torch.inference_mode() could not be updated in place by a later call that runs
            # under plain no_grad ("Inplace update to inference tensor outside InferenceMode").
            with torch.inference_mode(False), torch.no_grad():
                static_in = {k: _static_copy(k, sample[k]) for k in _KEYS if k in sample}
                static_in.update({f: True for f in _FLAGS if sample.get(f, False)})
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(self.warmup):          # populates the Prepared cache and the allocator pools
                        run(static_in)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                gc_was_on = gc.isenabled()
                gc.collect()
                gc.disable()                  # no collector run may free a stream / event / graph while the capture is open
                try:
                    if self.segmented:
                        global _SEGMENTER
                        # With M324_KV_OVERLAP=1 an exchange is IN FLIGHT while the next link of the chain is being captured, and the
                        # transport's own threads (gloo's copies through host memory; RCCL's watchdog) keep calling into HIP -- in
                        # "global" mode any such call invalidates the capture on this thread, so THAT form (and only that form)
                        # captures "thread_local" whatever the caller asked for.  Without the overlap the caller's mode stands.
                        from . import Pcd_motion
                        mode = self.capture_error_mode
                        if Pcd_motion.KV_OVERLAP and mode == "global":
                            mode = "thread_local"
                        g = _Segmenter(mode)
                        cap_stream = torch.cuda.Stream()
                        cap_stream.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(cap_stream):
                            _SEGMENTER = g
                            g.begin()
                            try:
                                static_out = run(static_in)
                            except BaseException:
                                _SEGMENTER = None
                                try:                      # close the open capture; an invalidated one raises again --
                                    g.end()               # the caller must see run()'s error, not this one
                                except Exception:
                                    pass
                                raise
                            _SEGMENTER = None
                            g.end()
                        torch.cuda.current_stream().wait_stream(cap_stream)
                    else:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, capture_error_mode=self.capture_error_mode):
                            static_out = run(static_in)
                finally:
                    if gc_was_on:
                        gc.enable()
            entry = (g, static_in, static_out)
        else:
            del self._graphs[key]                            # re-insert: most recently used last
        self._graphs[key] = entry
        g, static_in, static_out = entry
        with torch.inference_mode(False), torch.no_grad():
            for k, buf in static_in.items():
                if not isinstance(buf, torch.Tensor):
                    continue
                src = sample[k]
                if src.data_ptr() != buf.data_ptr():
                    buf.copy_(src, non_blocking=True)
        g.replay()
        out = edict(input_data=sample, pcd_moved=static_out["pcd_moved"])
        if "loss_metrics" in static_out:
            out.loss_metrics = static_out["loss_metrics"]
        if "reuse" in static_out:                    # static buffers like pcd_moved: overwritten by this graph's next replay
            out.reuse = static_out["reuse"]
        return out
