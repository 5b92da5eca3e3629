#!/usr/bin/env python3
"""Golden index maps of the reference's sliding-window driver.

Runs in the build container only.  The reference's `run_model_inference`
(scripts/inference_with_video_mesh.py:132-256) lives in a script whose module-level imports (bpy, trimesh,
imageio, ...) are not installable here, so the function's source is cut out of the file with `ast` and executed
on its own with a stub model: frame t of the video is filled with the value t and the stub returns, for every
input frame, that value -- the merged trajectory therefore spells out WHICH input frame produced each output
frame (and -1 where the driver overwrites frame 0 with ref_pcd).  Output: tests/golden/chunks.json.
"""
import ast
import json
import os

import torch

SRC = "/root/reference/scripts/inference_with_video_mesh.py"
HERE = os.path.dirname(os.path.abspath(__file__))

tree = ast.parse(open(SRC).read())
fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "run_model_inference")
ns = {"torch": torch, "print": lambda *a, **k: None}
exec(compile(ast.Module(body=[fn], type_ignores=[]), SRC, "exec"), ns)
run_model_inference = ns["run_model_inference"]


class Cfg(dict):
    __getattr__ = dict.__getitem__


def stub_model(sample):
    v = sample["rgb_video"]                      # [1, C, 1, 1, 3]
    vals = v[0, :, 0, 0, 0]                      # frame ids
    n = sample["ref_pcd"].shape[1]
    return {"pcd_moved": vals.view(1, -1, 1, 1).expand(1, -1, n, 3).clone()}


cases = {}
for C in (4, 8, 12, 32):
    for T in sorted({1, 2, C - 1, C, C + 1, 2 * C - 2, 2 * C - 1, 2 * C, 2 * C + 3, 3 * C - 2, 3 * C + 5, 30, 77, 256}):
        if T < 1:
            continue
        video = torch.arange(T, dtype=torch.float32).view(T, 1, 1, 1).expand(T, 1, 1, 3).contiguous()
        inp = {"ref_pcd": torch.full((1, 2, 3), -1.0)}
        cfg = Cfg(training=Cfg(frames=C, use_amp=False, amp_dtype="bf16"))
        out = run_model_inference(stub_model, inp, video, cfg, "cpu")
        cases[f"{T},{C}"] = None if out is None else [int(x) for x in out[0, :, 0, 0].tolist()]

json.dump(cases, open(os.path.join(HERE, "chunks.json"), "w"), indent=0, separators=(",", ":"))
print(len(cases), "cases;", "T=30,C=12 ->", cases["30,12"])
