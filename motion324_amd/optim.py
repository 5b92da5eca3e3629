"""Optimizer step and gradient plumbing of the reference's training loop on libm324 kernels.

Reference semantics reproduced (train.py:171-219, utils/training_utils.py:38-82):
  * AdamW(lr, betas=(beta1, beta2), eps 1e-8, weight_decay on parameters with dim() > 1 only, fused);
  * gradients: nan_to_num(nan=0, posinf=1e-6, neginf=-1e-6), global L2 norm, clip to grad_clip_norm,
    skip the update when the pre-clip norm exceeds allowed_gradnorm_factor * grad_clip_norm or the loss is not finite;
  * cosine schedule with linear warm-up (transformers.get_cosine_schedule_with_warmup).

Memory layout.  Parameters, gradients and both Adam moments live in FOUR flat fp32 buffers with one common layout
(628 MB each for the full model): the parameters that take weight decay first (training_utils.py:39-47's first
group), the 1-D ones after them, every tensor starting on a 16-byte boundary.  The update is then ONE launch of
m324_adamw_flat over the whole buffer (multi-tensor semantics of torch's fused AdamW) and the model's nn.Parameters
are views into the flat parameter buffer (``flatten=True``; same trick as apex / DDP's flat buckets).

Data parallelism (train.py:88-89,159-166: DDP's reducer).  Inside each group the tensors are ordered by the moment
their gradient is complete in training.forward_backward (decoder first, the point embedding last), so the flat
gradient buffer fills front to back and is cut into buckets of ~`bucket_mb` MB.  As soon as the backward has
finished every tensor of a bucket (GradStore.done -> FusedAdamW.ready) the bucket's all-reduce is launched on a side
stream: RCCL moves it over xGMI while the compute stream keeps running the backward of the earlier blocks.
`finish_reduce()` joins before the gradient norm.  xGMI is point-to-point, so the buckets are few and large (50 MB
default, ~13 for the full model) rather than DDP's 25 MiB.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import lib as L
from . import ops, parallel
from . import prepared, switches
from .prepared import bump_generation


WEIGHT_MIRROR = switches.flag("M324_WEIGHT_MIRROR")


def cosine_with_warmup(step: int, warmup: int, total: int, base_lr: float) -> float:
    """lr at optimizer step `step` (0-based), as transformers.get_cosine_schedule_with_warmup(num_cycles=0.5)."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


def backward_completion_order(model) -> List[torch.nn.Parameter]:
    """Trainable parameters of a Motion_Latent_Model in the order training.forward_backward finishes their gradients
    (the GradStore.done calls there).  Anything the walk does not know is appended at the end."""
    order: List[torch.nn.Parameter] = []
    seen = set()

    def take(ps: Iterable[torch.nn.Parameter]):
        for p in ps:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                order.append(p)

    g = lambda name: getattr(model, name, None)
    if g("decoder_cross_attn") is not None:
        take(model.shared_mlp_output.parameters())
        take(model.decoder_cross_attn.parameters())
        for gb, lb in reversed(list(zip(model.global_transformer_blocks, model.local_transformer_blocks))):
            take(lb.parameters())
            take(gb.parameters())
        take(model.transformer_input_layernorm.parameters())
        take([model.special_token_0, model.special_token_rest])
        for b in reversed(list(model.points_transformer_blocks)):
            take(b.parameters())
        take(model.encoder_cross_attn.parameters())
        take([model.learnable_tokens])
        take(model.point_normal_rgb_proj.parameters())
        take(model.point_embed.parameters())
    take(model.parameters())
    return order


class FusedAdamW:
    """AdamW + gradient clipping / skipping + (bucketed, overlapped) data-parallel gradient averaging.

    named_params: iterable of (name, parameter) -- e.g. model.named_parameters(); `order` (optional) lists the
    parameters in gradient-completion order (backward_completion_order(model)) so that buckets can leave early."""

    def __init__(self, named_params, lr=4e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, grad_clip_norm=1.0,
                 allowed_gradnorm_factor=5.0, group=None, order: Optional[Sequence[torch.nn.Parameter]] = None,
                 flatten: bool = True, bucket_mb: float = 50.0):
        named = [(n, p) for n, p in named_params if p.requires_grad]
        if not named:
            raise ValueError("FusedAdamW: no trainable parameters")
        rank_of = {id(p): i for i, p in enumerate(order)} if order is not None else {}
        key = lambda np_: rank_of.get(id(np_[1]), len(rank_of))
        decay = sorted([np_ for np_ in named if np_[1].dim() > 1 and not getattr(np_[1], "_no_weight_decay", False)], key=key)
        nodecay = sorted([np_ for np_ in named if not (np_[1].dim() > 1 and not getattr(np_[1], "_no_weight_decay", False))], key=key)
        self.names: List[str] = [n for n, _ in decay + nodecay]
        self.params: List[torch.nn.Parameter] = [p for _, p in decay + nodecay]
        self.n_decay_params = len(decay)
        # Checkpoint numbering = the reference's create_optimizer (utils/training_utils.py:38-52): named_parameters()
        # order, the decay group first.  The LAYOUT above may be re-sorted by gradient-completion order (buckets); the
        # numbering a checkpoint carries never is.  ckpt_names[k] = name of checkpoint parameter k.
        is_decay = lambda p: p.dim() > 1 and not getattr(p, "_no_weight_decay", False)
        self.ckpt_names: List[str] = [n for n, p in named if is_decay(p)] + [n for n, p in named if not is_decay(p)]
        self.decay = [i < self.n_decay_params for i in range(len(self.params))]
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.clip, self.skip_factor, self.group = grad_clip_norm, allowed_gradnorm_factor, group
        dev = self.params[0].device
        if dev.type != "cuda":
            raise L.M324Error("FusedAdamW needs the parameters on a HIP device (model.to('cuda') first)")
        # layout: every tensor on a 16-byte boundary, the decay segment padded to a multiple of 4 elements
        self.offsets, o = [], 0
        for i, p in enumerate(self.params):
            if i == self.n_decay_params:
                self.n_decay = o
            self.offsets.append(o)
            o += (p.numel() + 3) // 4 * 4
        if self.n_decay_params == len(self.params):
            self.n_decay = o
        self.numel = o
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self.flat_param = torch.zeros(o, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(o, dtype=torch.float32, device=dev)
        self.m = torch.zeros(o, dtype=torch.float32, device=dev)
        self.v = torch.zeros(o, dtype=torch.float32, device=dev)
        self.flatten = flatten
        with torch.no_grad():
            for i, p in enumerate(self.params):
                view = self.flat_param[self.offsets[i]:self.offsets[i] + p.numel()].view(p.shape)
                view.copy_(p.data)
                if flatten:
                    p.data = view              # the parameter now lives in the flat buffer
        if flatten:
            bump_generation()                  # parameter storage moved: drop cached kernel-ready copies
        self._mirror_params: List[torch.nn.Parameter] = []
        if flatten and WEIGHT_MIRROR:
            self._build_mirror(dev)
        self.step_count = 0
        self._partial = torch.empty(1024, dtype=torch.float32, device=dev)
        self._sumsq = torch.zeros((), dtype=torch.float32, device=dev)
        self._gscale = torch.ones((), dtype=torch.float32, device=dev)
        # buckets: contiguous element ranges of the flat gradient, in layout (= completion) order
        self.buckets: List[Tuple[int, int, List[int]]] = []        # (start, end, parameter indices)
        cap = max(1, int(bucket_mb * 1e6 / 4))
        start, members = 0, []
        for i, p in enumerate(self.params):
            members.append(i)
            end = self.offsets[i] + (p.numel() + 3) // 4 * 4
            if end - start >= cap or i == len(self.params) - 1:
                self.buckets.append((start, end, members))
                start, members = end, []
        self._bucket_of = {i: b for b, (_, _, mem) in enumerate(self.buckets) for i in mem}
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self._reset_reduce_state()

    # ------------------------------------------------------------------ bf16 weight mirror
    def _build_mirror(self, dev: torch.device) -> None:
        """bf16 copies of every Linear weight of the flat parameter buffer -- [N, K] for the forward GEMMs, [K, round_up(N, 64)] for
        the dgrad GEMMs -- in two flat buffers that ONE m324_weight_mirror launch rewrites after each update (instead of ~115 torch
        casts and ~90 m324_transpose launches per step: Prepared hands the views out while they are current, prepared.register_mirror).
        Weights whose K is not a multiple of the GEMM's K-tile (the 51- and 774-wide point embeddings) stay on Prepared's padding path."""
        items, o, ot, tiles = [], 0, 0, 0
        for i, p in enumerate(self.params):
            if p.dim() != 2:              # Linear weights only (the token tables are not GEMM operands)
                continue
            n, k = p.shape
            if k % prepared.K_ALIGN != 0:
                continue
            ldt = (n + 63) // 64 * 64
            items.append((i, self.offsets[i], o, ot, tiles, n, k, ldt))
            o += (n * k + 63) // 64 * 64
            ot += k * ldt
            tiles += ((n + 63) // 64) * ((k + 63) // 64)
        if not items:
            return
        self._mirror = torch.empty(o, dtype=torch.bfloat16, device=dev)
        self._mirror_t = torch.zeros(ot, dtype=torch.bfloat16, device=dev)          # the pad columns of the transposed copies stay zero
        arr = (L.MirrorItem * len(items))()
        for j, (i, so, do, dto, ft, n, k, ldt) in enumerate(items):
            arr[j].src_off, arr[j].dst_off, arr[j].dstT_off, arr[j].first_tile = so, do, dto, ft
            arr[j].rows, arr[j].cols, arr[j].ldT = n, k, ldt
            p = self.params[i]
            prepared.register_mirror(p, self._mirror[do:do + n * k].view(n, k), self._mirror_t[dto:dto + k * ldt].view(k, ldt), self._mirror, do)
            self._mirror_params.append(p)
        self._mirror_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self._mirror_n, self._mirror_tiles = len(items), tiles
        self.sync_mirror()

    def sync_mirror(self) -> None:
        """Rewrites the bf16 weight copies from the flat parameters as they are now (after an update, or after anything else wrote
        the parameters: load_state_dict, a manual edit) and vouches for them."""
        if not self._mirror_params:
            return
        ops.weight_mirror(self.flat_param, self._mirror, self._mirror_t, self._mirror_table, self._mirror_n, self._mirror_tiles)
        prepared.validate_mirrors(self._mirror_params)

    # ------------------------------------------------------------------ gradient buffer
    def owns(self, p) -> bool:
        return id(p) in self._index

    def grad_view(self, i: int) -> torch.Tensor:
        p = self.params[i]
        return self.flat_grad[self.offsets[i]:self.offsets[i] + p.numel()]

    def grad_of(self, p: torch.nn.Parameter) -> torch.Tensor:
        return self.grad_view(self._index[id(p)]).view(p.shape)

    def load_grads(self, G, accumulate: bool = False) -> None:
        """Copies a GradStore (or anything with .get(param)) into the flat buffer (torch copies: data movement only).
        Not needed when the GradStore was created with ``sink=optimizer``: it then writes into the buffer itself."""
        if getattr(G, "sink", None) is self:
            return
        for i, p in enumerate(self.params):
            g = G.get(p)
            if g is None:
                if not accumulate:
                    self.grad_view(i).zero_()
                continue
            if accumulate:
                self.grad_view(i).add_(g.reshape(-1))
            else:
                self.grad_view(i).copy_(g.reshape(-1))

    # ------------------------------------------------------------------ data-parallel averaging
    def _world(self) -> int:
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def _reset_reduce_state(self) -> None:
        self._ready = [False] * len(self.params)
        self._pending = [len(mem) for _, _, mem in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works: List = []
        self.launch_log: List[Tuple[int, int]] = []     # (bucket, id of the stream it was enqueued on): tests / traces

    def begin_step(self) -> None:
        """Call before a backward whose GradStore uses this optimizer as its sink."""
        self._reset_reduce_state()

    def step_launches(self) -> int:
        """Gradient buckets of the current step already handed to the other ranks (a step cannot be redone after that)."""
        return len(self._works)

    def _launch_bucket(self, b: int) -> None:
        world = self._world()
        self._launched[b] = True
        if not parallel.collectives_on(world):
            return
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.flat_grad.device)
        s0, s1, _ = self.buckets[b]
        chunk = self.flat_grad[s0:s1]
        comm = self._comm_stream
        comm.wait_stream(torch.cuda.current_stream())          # the bucket's gradients are complete on the compute stream
        with torch.cuda.stream(comm):
            self.launch_log.append((b, comm.cuda_stream))
            work = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((work, chunk))

    def ready(self, params: Iterable[torch.nn.Parameter]) -> None:
        """The backward will not touch these parameters' gradients again: launch every bucket that became complete."""
        for p in params:
            i = self._index.get(id(p))
            if i is None or self._ready[i]:
                continue
            self._ready[i] = True
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0 and not self._launched[b]:
                self._launch_bucket(b)

    def finish_reduce(self) -> None:
        """Launches whatever bucket is still waiting, then makes the compute stream wait for all of them and turns the
        sums into means."""
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch_bucket(b)
        world = self._world()
        if self._works:
            comm = self._comm_stream
            with torch.cuda.stream(comm):
                for work, chunk in self._works:
                    work.wait()                                # orders the collective before the scaling on `comm`
                    chunk.mul_(1.0 / world)
            torch.cuda.current_stream().wait_stream(comm)
        self._works = []

    def all_reduce_mean(self) -> None:
        """Blocking form (no overlap): every bucket now.  Kept for callers that filled flat_grad with load_grads()."""
        self._reset_reduce_state()
        self.finish_reduce()

    # ------------------------------------------------------------------ update
    def step(self, lr: Optional[float] = None, loss_is_finite: bool = True) -> Dict[str, float]:
        """One optimizer update from self.flat_grad.  Returns {'grad_norm', 'skipped'} (one host sync, as the
        reference's clip_grad_norm_(...).item())."""
        ops.grad_sumsq(self.flat_grad, self._sumsq, self._partial, sanitize=True, accumulate=False)
        norm = float(self._sumsq.sqrt())
        skipped = (not loss_is_finite) or (not math.isfinite(norm)) or norm > self.skip_factor * self.clip
        if not skipped:
            self.step_count += 1
            self._gscale.fill_(min(1.0, self.clip / (norm + 1e-6)))
            lr = self.lr if lr is None else lr
            if self.flatten:
                ops.adamw_flat(self.flat_param, self.flat_grad, self.m, self.v, self.n_decay, lr, self.betas[0], self.betas[1],
                               self.eps, self.wd, self.step_count, self._gscale)
            else:
                for i, p in enumerate(self.params):
                    o, n = self.offsets[i], p.numel()
                    ops.adamw_step(p.data.view(-1), self.flat_grad[o:o + n], self.m[o:o + n], self.v[o:o + n], lr, self.betas[0],
                                   self.betas[1], self.eps, self.wd if self.decay[i] else 0.0, self.step_count, self._gscale)
            bump_generation()           # the kernel wrote parameter memory behind torch's back: drop cached bf16 copies
            if self.flatten:
                self.sync_mirror()      # ... and rewrite the ones this optimizer keeps itself
        return {"grad_norm": norm, "skipped": skipped}

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.flat_grad.zero_()

    # ------------------------------------------------------------------ checkpointing (torch.optim.AdamW's layout)
    def state_dict(self) -> dict:
        """{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [decay group, no-decay group]} with the
        parameter numbering of utils/training_utils.py:38-52 (decay parameters first), so the dict loads into
        torch.optim.AdamW built by the reference's create_optimizer and vice versa."""
        state = {}
        layout = {n: i for i, n in enumerate(self.names)}
        if self.step_count > 0:
            for k, name in enumerate(self.ckpt_names):       # k: the reference's parameter number, i: our layout slot
                i = layout[name]
                p = self.params[i]
                o, n = self.offsets[i], p.numel()
                state[k] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.m[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.v[o:o + n].view(p.shape).clone()}
        common = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "amsgrad": False, "maximize": False,
                  "foreach": None, "capturable": False, "differentiable": False, "fused": True,
                  "decoupled_weight_decay": True}
        nd = self.n_decay_params
        groups = [dict(common, weight_decay=self.wd, params=list(range(nd))),
                  dict(common, weight_decay=0.0, params=list(range(nd, len(self.params))))]
        return {"state": state, "param_groups": groups, "param_names": list(self.ckpt_names)}

    def load_state_dict(self, sd: dict) -> None:
        groups = sd.get("param_groups", [])
        ids = [i for g in groups for i in g.get("params", [])]
        if ids and len(ids) != len(self.params):
            raise ValueError(f"optimizer state has {len(ids)} parameters, this optimizer {len(self.params)}")
        # A dict without names comes from torch.optim.AdamW as the reference builds it (torch drops unknown keys and
        # never writes names): its numbering is create_optimizer's = self.ckpt_names.  Ours carries the names anyway.
        names = sd.get("param_names")
        names = list(self.ckpt_names) if names is None else list(names)
        pos = {n: k for k, n in enumerate(names)}
        if set(pos) != set(self.names):
            raise ValueError("optimizer state names do not match the parameters")
        remap = [pos[n] for n in self.names]               # layout slot i <- checkpoint parameter remap[i]
        state = sd.get("state", {})
        steps = set()
        self.m.zero_()
        self.v.zero_()
        for i, p in enumerate(self.params):
            st = state.get(remap[i], state.get(str(remap[i])))
            if st is None:
                continue
            o, n = self.offsets[i], p.numel()
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state {self.names[i]}: shape {tuple(st['exp_avg'].shape)} vs {tuple(p.shape)}")
            self.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"optimizer state carries different step counts {sorted(steps)}")
        self.step_count = steps.pop() if steps else 0
        if groups:
            g0 = groups[0]
            self.lr = g0.get("lr", self.lr)
            self.betas = tuple(g0.get("betas", self.betas))
            self.eps = g0.get("eps", self.eps)
            self.wd = g0.get("weight_decay", self.wd)
