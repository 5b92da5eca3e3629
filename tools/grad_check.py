#!/usr/bin/env python3
"""Per-module finiteness / norm report of one training step's gradients at the dyscene.yaml shapes
(B=.. PREC=bf16|fp32 environment variables).  Found the LDS-DMA race in the attention backward (round 1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motion324_amd as m
from motion324_amd import synth, training
dev = torch.device("cuda")
cfg = synth.make_config(frames=12)
model = m.Motion_Latent_Model(cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.Dims(frames=12), seed=0).items()}, strict=False)
model = model.train().to(dev)
B = int(os.environ.get("B", 8))
s = synth.synth_inputs(B, 12, 4096, 4096, 224, seed=1, with_target=True)
sample = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
m.set_precision(os.environ.get("PREC", "bf16"))
for it in range(3):
    loss, _, G = training.forward_backward(model, sample)
    torch.cuda.synchronize()
    bad = []
    tot = 0.0
    for name, p in model.named_parameters():
        if not p.requires_grad: continue
        g = G.get(p)
        n = float(g.double().norm())
        fin = bool(torch.isfinite(g).all())
        tot += n * n if fin else float("inf")
        if not fin or n > 1e3: bad.append((name, n, fin, float(g.abs().max())))
    import collections
    st = collections.OrderedDict()
    for name, p in model.named_parameters():
        if not p.requires_grad: continue
        g = G.get(p)
        key = ".".join(name.split(".")[:2]) if name.split(".")[0].endswith("blocks") else name.split(".")[0]
        a = st.setdefault(key, [0, 0])
        a[0] += 1; a[1] += int(bool(torch.isfinite(g).all()))
    print("iter", it, "loss", float(loss), "total norm", tot ** 0.5, {k: f"{v[1]}/{v[0]}" for k, v in st.items()}, flush=True)
    break
