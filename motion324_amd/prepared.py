"""Kernel-ready views of a module's parameters (compute-dtype copies, K padding), cached.

libm324's GEMM wants K-major weight rows in the compute dtype with K a multiple of the K-tile; the
parameters themselves stay fp32 nn.Parameters with the reference's names and shapes.  A Prepared
object converts each parameter once per (device, dtype) and re-converts it when the parameter is
modified in place (optimizer step, load_state_dict) or re-allocated (``.to(device)``).
This is one-time weight preparation, done with torch copies; no arithmetic of the path lives here.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, Optional, Sequence, Tuple

import torch

K_ALIGN = 64   # GEMM K-tile of the bf16 kernel (fp32 needs 32); one padding rule for both modes

_OVERRIDE: Optional[torch.dtype] = None


def set_precision(p: Optional[str]) -> None:
    """Force 'bf16' / 'fp32' for every forward, or None to follow torch.autocast (the reference's switch)."""
    global _OVERRIDE
    _OVERRIDE = {None: None, "bf16": torch.bfloat16, "fp32": torch.float32}[p]


def compute_dtype() -> torch.dtype:
    """bf16 speed mode under torch.autocast(cuda, bf16/fp16) -- what the reference's callers request
    (train.py:150-155, scripts/inference_with_video_mesh.py:162-167) -- else fp32 parity mode."""
    if _OVERRIDE is not None:
        return _OVERRIDE
    env = os.environ.get("M324_PRECISION")
    if env:
        return {"bf16": torch.bfloat16, "fp32": torch.float32}[env]
    if torch.is_autocast_enabled("cuda"):
        return torch.bfloat16      # fp16 autocast is served by the bf16 kernels (same operand width)
    return torch.float32


_GENERATION = 0


def bump_generation() -> None:
    """Invalidates every cached kernel-ready copy.  Call after updating parameters through raw pointers (the fused
    AdamW kernel writes parameter memory without touching torch's version counters; so does
    torch.optim.AdamW(fused=True), which is why Motion_Latent_Model bumps it on every training-mode forward)."""
    global _GENERATION
    _GENERATION += 1


def generation() -> int:
    return _GENERATION


def weight_stamp(module: torch.nn.Module) -> int:
    """Cheap fingerprint of a module's parameter / buffer storage and in-place version counters: changes on
    load_state_dict, .to(device), an eager optimizer step or any other tracked in-place update (a hipGraph captured
    from the old weights must not be replayed after that)."""
    h = 0
    for t in list(module.parameters()) + list(module.buffers()):
        h = (h * 1000003 + t.data_ptr() * 31 + t._version) & 0xFFFFFFFFFFFFFFFF
    return h


class _Mirror:
    __slots__ = ("ref", "mat", "mat_t", "gen", "ptr", "version", "base", "off")


_MIRRORS: Dict[int, _Mirror] = {}


def register_mirror(p: torch.Tensor, mat: torch.Tensor, mat_t: Optional[torch.Tensor], base: Optional[torch.Tensor] = None, off: int = 0) -> None:
    """An optimizer that keeps bf16 copies of its parameters current itself (optim.FusedAdamW: one m324_weight_mirror launch per step
    over its flat parameter buffer) registers them here: mat [N, K] row-major and mat_t [K, round_up(N, 64)] for parameter p.  A copy
    is handed out by Prepared.mat / mat_t while validate_mirrors() has vouched for it SINCE the last change Prepared would notice
    (generation counter, the tensor's in-place version, its storage address); otherwise Prepared converts as for any parameter.
    base / off: the flat bf16 buffer mat is a view of and mat's first element in it (Prepared.cat_rows hands out ONE view over
    neighbouring weights -- the k and v projections of a cross-attention -- instead of concatenating copies)."""
    e = _Mirror()
    e.base, e.off = base, off
    e.ref, e.mat, e.mat_t, e.gen, e.ptr, e.version = weakref.ref(p, lambda _r, k=id(p): _MIRRORS.pop(k, None)), mat, mat_t, -1, 0, -1
    _MIRRORS[id(p)] = e


def validate_mirrors(ps: Sequence[torch.Tensor]) -> None:
    """The registered copies of these parameters were (re)written from their current values just now."""
    for p in ps:
        e = _MIRRORS.get(id(p))
        if e is not None and e.ref() is p:
            e.gen, e.ptr, e.version = _GENERATION, p.data_ptr(), p._version


def drop_mirrors(ps: Sequence[torch.Tensor]) -> None:
    for p in ps:
        _MIRRORS.pop(id(p), None)


def _mirror(p: torch.Tensor, dtype: torch.dtype, device: torch.device) -> Optional[_Mirror]:
    e = _MIRRORS.get(id(p))
    device = torch.device(device)
    if e is None or dtype != torch.bfloat16 or e.gen != _GENERATION or e.ref() is not p or e.ptr != p.data_ptr() or e.version != p._version \
            or e.mat.device.type != device.type or (device.index is not None and e.mat.device.index != device.index):
        return None
    return e


def pad_k(k: int) -> int:
    return (k + K_ALIGN - 1) // K_ALIGN * K_ALIGN


class Prepared:
    _registry: "weakref.WeakKeyDictionary[torch.nn.Module, Dict[Tuple[str, torch.dtype], Prepared]]" = \
        weakref.WeakKeyDictionary()

    def __init__(self, device: torch.device, dtype: torch.dtype):
        self.device, self.dtype = device, dtype
        self._cache: Dict[tuple, Tuple[tuple, torch.Tensor]] = {}

    @classmethod
    def for_module(cls, module: torch.nn.Module, device: torch.device, dtype: Optional[torch.dtype] = None) -> "Prepared":
        dtype = compute_dtype() if dtype is None else dtype
        per_mod = cls._registry.setdefault(module, {})
        key = (str(device), dtype)
        if key not in per_mod:
            per_mod[key] = cls(device, dtype)
        return per_mod[key]

    @staticmethod
    def _stamp(ps: Sequence[torch.Tensor]) -> tuple:
        # the generation counter guards against raw-pointer updates by an optimizer: frozen tensors (the 86 M DINOv2
        # weights, buffers) are never updated that way, so their copies survive a bump (no re-conversion per step)
        gen = _GENERATION if any(getattr(p, "requires_grad", False) for p in ps) else 0
        return (gen,) + tuple((p.data_ptr(), p._version) for p in ps)

    def _get(self, kind: str, ps: Sequence[torch.Tensor], make):
        key = (kind,) + tuple(id(p) for p in ps)
        stamp = self._stamp(ps)
        hit = self._cache.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        with torch.no_grad():
            t = make()
        self._cache[key] = (stamp, t)
        return t

    def _mat_of(self, p: torch.Tensor) -> torch.Tensor:
        w = p.detach().reshape(p.shape[0], -1)
        k = w.shape[1]
        kp = pad_k(k)
        if kp == k:
            return w.to(device=self.device, dtype=self.dtype).contiguous()
        out = torch.zeros((w.shape[0], kp), dtype=self.dtype, device=self.device)
        out[:, :k] = w
        return out

    def mat(self, p: torch.Tensor) -> torch.Tensor:
        """[N, K'] compute-dtype GEMM operand of a Linear/Conv weight (flattened, K zero-padded to 64)."""
        e = _mirror(p, self.dtype, self.device) if _MIRRORS else None
        if e is not None:
            return e.mat
        return self._get("mat", (p,), lambda: self._mat_of(p))

    def mat_t(self, p: torch.Tensor, transpose) -> torch.Tensor:
        """[K', round_up(N, 64)] transposed operand (the dgrad GEMMs'): the optimizer's mirror when it is current, else
        transpose(self.mat(p)), cached like mat."""
        e = _mirror(p, self.dtype, self.device) if _MIRRORS else None
        if e is not None and e.mat_t is not None:
            return e.mat_t
        return self.derived("matT", (p,), lambda: transpose(self.mat(p)))

    def cat_rows(self, ps: Sequence[torch.Tensor]) -> torch.Tensor:
        """Several [N_i, K] weights stacked along N (one GEMM for k and v projections)."""
        es = [_mirror(p, self.dtype, self.device) for p in ps] if _MIRRORS else [None]
        if all(e is not None and e.base is not None for e in es) and all(e.base is es[0].base and e.mat.shape[1] == es[0].mat.shape[1] for e in es) \
                and all(es[i].off + es[i].mat.numel() == es[i + 1].off for i in range(len(es) - 1)):
            k = es[0].mat.shape[1]
            return es[0].base[es[0].off:es[-1].off + es[-1].mat.numel()].view(-1, k)
        return self._get("cat", tuple(ps), lambda: torch.cat([self._mat_of(p) for p in ps], dim=0).contiguous())

    def vec(self, p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """fp32 epilogue / normalisation vector (bias, LN weight, LayerScale gamma ...)."""
        if p is None:
            return None
        if p.dtype == torch.float32 and p.device == self.device and p.is_contiguous():
            return p.detach().reshape(-1)
        return self._get("vec", (p,), lambda: p.detach().to(device=self.device, dtype=torch.float32).contiguous().reshape(-1))

    def cat_vecs(self, ps: Sequence[Optional[torch.Tensor]]) -> Optional[torch.Tensor]:
        if all(p is None for p in ps):
            return None
        if any(p is None for p in ps):
            raise ValueError("cat_vecs: either all or none of the biases must exist")
        return self._get("catv", tuple(ps), lambda: torch.cat(
            [p.detach().to(device=self.device, dtype=torch.float32).reshape(-1) for p in ps]).contiguous())

    def folded(self, ln_w: torch.Tensor, ln_b: Optional[torch.Tensor], w: torch.Tensor, b: Optional[torch.Tensor]):
        """A LayerNorm folded into the Linear behind it (m324_gemm's LayerNorm fold): returns
        (W' bf16 [N, K] = w_ln[k] W[n, k], colsum fp32 [N] = sum_k W'[n, k] of the ROUNDED values, bias' fp32 [N] =
        b + W b_ln or None).  One-time weight preparation, like mat()."""
        ps = tuple(p for p in (ln_w, ln_b, w, b) if p is not None)

        def make():
            W = w.detach().reshape(w.shape[0], -1).to(device=self.device, dtype=torch.float32)
            if W.shape[1] % K_ALIGN:
                raise ValueError("folded(): K must be a multiple of 64")
            Wf = (W * ln_w.detach().to(device=self.device, dtype=torch.float32)[None, :]).to(torch.bfloat16).contiguous()
            colsum = Wf.double().sum(dim=1).float().contiguous()
            bias = None
            if ln_b is not None or b is not None:
                bias = torch.zeros(W.shape[0], dtype=torch.float64, device=self.device)
                if b is not None:
                    bias += b.detach().to(device=self.device, dtype=torch.float64)
                if ln_b is not None:
                    bias += W.double() @ ln_b.detach().to(device=self.device, dtype=torch.float64)
                bias = bias.float().contiguous()
            return (Wf, colsum, bias)
        return self._get("folded", ps, make)

    def f32(self, p: torch.Tensor) -> torch.Tensor:
        """fp32 contiguous device copy of a parameter/buffer of any shape (tokens, position tables)."""
        if p.dtype == torch.float32 and p.device == self.device and p.is_contiguous():
            return p.detach()
        return self._get("f32", (p,), lambda: p.detach().to(device=self.device, dtype=torch.float32).contiguous())

    def derived(self, name: str, ps: Sequence[torch.Tensor], make) -> torch.Tensor:
        """Cached tensor derived from parameters by ``make()`` (e.g. an interpolated position table)."""
        return self._get("derived:" + name, tuple(ps), make)
