#!/usr/bin/env python3
"""Golden of the caller's pre-step, produced by executing the reference's OWN functions --
`prepare_mesh_data` (scripts/inference_with_video_mesh.py:60-129) and `sample_pointcloud_with_albedo`
(utils/mesh_processing.py:130-191), cut out of their files with `ast` -- on a stand-in mesh object.

trimesh is not installable offline, so `trimesh.load` returns a small in-memory mesh (random closed surface: a
perturbed icosphere) whose `.sample()` is the repo's deterministic sampler; everything downstream of the sampler --
unit-cube normalisation in the reference's two precisions, face normals at the samples, vertex-colour averaging, the
cKDTree nearest-sample colours, dtypes / shapes / keys of the packed dict -- is the reference's code.
Build container only.  Output: tests/golden/prestep.npz (inputs + expected outputs)."""
import ast
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from motion324_amd import preprocess, synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def icosphere(level=2):
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(level):
        cache, nf = {}, []
        v = list(v)

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = (v[a] + v[b]) / 2
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(v), np.array(nf, dtype=np.int64)
    return v, f


class StubVisual:
    def __init__(self, vc):
        self.vertex_colors = vc


class StubMesh:
    def __init__(self, vertices, faces, colors, seed):
        self.vertices = vertices
        self.faces = faces
        self.visual = StubVisual(colors)
        self._seed = seed

    def fix_normals(self):
        pass

    @property
    def vertex_normals(self):           # area-weighted mean of the incident face normals
        fn = preprocess.face_normals(self.vertices, self.faces)
        tri = np.asarray(self.vertices)[self.faces]
        area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1, keepdims=True)
        out = np.zeros_like(np.asarray(self.vertices, dtype=np.float64))
        for k in range(3):
            np.add.at(out, self.faces[:, k], fn * area)
        return out / np.maximum(np.linalg.norm(out, axis=1, keepdims=True), 1e-30)

    @property
    def face_normals(self):
        return preprocess.face_normals(self.vertices, self.faces)

    @property
    def triangles(self):
        return np.asarray(self.vertices)[self.faces]

    def sample(self, num, return_index=False):
        pts, fi = preprocess.sample_surface(self.vertices, self.faces, num, self._seed)
        return (pts, fi) if return_index else pts


def extract(path, names, ns):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)


v, f = icosphere(2)
v = v * np.array([1.7, 0.9, 1.2]) + 0.15 * synth.normal(3, "bump", v.shape) + np.array([3.0, -2.0, 0.5])    # off-centre blob
colors = (synth.uniform(3, "vc", (len(v), 4)) * 255).astype(np.uint8)
mesh = StubMesh(v.copy(), f, colors, seed=11)

trimesh = types.ModuleType("trimesh")
trimesh.Scene = type("Scene", (), {})
trimesh.load = lambda path, force=None: mesh
trimesh.util = types.SimpleNamespace(concatenate=lambda geoms: None)
sys.modules["trimesh"] = trimesh

ns = {"np": np, "torch": torch, "print": lambda *a, **k: None, "Image": None}
extract("/root/reference/utils/mesh_processing.py", {"sample_pointcloud_with_albedo", "barycentric_coords", "normalize_mesh"}, ns)
extract("/root/reference/scripts/inference_with_video_mesh.py", {"prepare_mesh_data"}, ns)


class Cfg(dict):
    __getattr__ = dict.__getitem__


NUM = 3000
vn_before = mesh.vertex_normals.astype(np.float32)
inp, mesh_out, faces = ns["prepare_mesh_data"](Cfg(training=Cfg(num_shape_samples=NUM)), "stub.glb", "cpu")
save = {"in_vertices": v, "in_faces": f, "in_vertex_colors": colors, "in_vertex_normals": vn_before,
        "num_shape_samples": np.int64(NUM), "seed": np.int64(11), "mesh_vertices_out": np.asarray(mesh_out.vertices)}
for k, t in inp.items():
    save["out_" + k] = t.numpy()
# normalize_mesh (return_params=True) on the raw vertices
nv, center, scale = ns["normalize_mesh"](StubMesh(v.copy(), f, colors, 0), return_params=True)
save.update(norm_vertices=nv, norm_center=center, norm_scale=np.float32(scale))
np.savez_compressed(os.path.join(HERE, "prestep.npz"), **save)
print({k: (a.shape, a.dtype) for k, a in save.items()})
