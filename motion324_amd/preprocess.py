"""Pre-step of the path: mesh -> the tensors Motion_Latent_Model.forward consumes.

Counterpart of the reference caller's `prepare_mesh_data` (scripts/inference_with_video_mesh.py:60-129) with
`normalize_mesh` / `sample_pointcloud_with_albedo` (utils/mesh_processing.py:130-191,194-...): unit-cube
normalisation of the vertices, area-weighted surface sampling with per-sample normal and colour, nearest-sample colour
for every vertex, and the packing into `ref_shape_pcd / ref_shape_normals / ref_shape_rgbs / ref_pcd / ref_normal /
ref_rgb / faces` ([1, n, 3] fp32, faces int64) on the device.

The reference reads the mesh with trimesh (not installable offline) and draws the samples from trimesh's sampler with
numpy's global RNG; here the mesh arrives as plain arrays and the sampler is a repo-owned counter-based one
(motion324_amd.synth's SplitMix64 stream), so a mesh + seed gives the same samples on every machine.  The arithmetic
that IS numpy in the reference -- normalisation, vertex-colour averaging, nearest-sample colours, packing -- is pinned
against the reference's own code run on a stand-in mesh (tests/golden/make_prestep_golden.py).  The nearest-sample
search is the HIP kernel m324_nearest_point (a cKDTree query on the CPU in the reference).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import ops, synth


def normalize_vertices(vertices: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.float32]:
    """float32 vertices -> (vertices in [-0.5, 0.5]^3, center, scale = 2 (max |v - center| + 1e-8))
    (utils/mesh_processing.py normalize_mesh; scripts/inference_with_video_mesh.py:93-97)."""
    v = np.asarray(vertices).astype(np.float32)
    center = (v.max(axis=0) + v.min(axis=0)) / 2
    v = v - center
    v_max = np.abs(v).max()
    scale = 2 * (v_max + 1e-8)
    return v / scale, center, scale


def face_normals(vertices: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Unit normals of the triangles (zero for degenerate ones), float64 like trimesh's."""
    tri = np.asarray(vertices, dtype=np.float64)[faces]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return np.where(ln > 0, n / np.maximum(ln, 1e-300), 0.0)


def sample_surface(vertices: np.ndarray, faces: np.ndarray, num: int, seed: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """`num` points distributed uniformly over the surface: faces drawn with probability proportional to their area,
    uniform barycentric coordinates by reflection (the published construction trimesh.sample implements).  Returns
    (points float64 [num, 3], face index int64 [num]); deterministic in (mesh, num, seed)."""
    tri = np.asarray(vertices, dtype=np.float64)[faces]
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    area = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
    cum = np.cumsum(area)
    u = synth.uniform(seed, "surface.face", (num,)).astype(np.float64)
    fi = np.minimum(np.searchsorted(cum, u * cum[-1], side="right"), len(faces) - 1).astype(np.int64)
    r = synth.uniform(seed, "surface.bary", (num, 2)).astype(np.float64)
    flip = r.sum(axis=1) > 1.0
    r[flip] = 1.0 - r[flip]
    pts = tri[fi, 0] + r[:, :1] * e1[fi] + r[:, 1:] * e2[fi]
    return pts, fi


def sample_pointcloud_with_albedo(vertices, faces, num: int, vertex_colors: Optional[np.ndarray] = None, seed: int = 0):
    """(points, normals, colours) float32 [num, 3] (utils/mesh_processing.py:130-191): face normals at the samples; the
    colour of a sample is the mean of its triangle's vertex colours (uint8 RGB(A) / 255) when the mesh has them, else 0.5
    grey.  (The reference's third source, a UV texture lookup, needs the image file and is the caller's business.)"""
    pts, fi = sample_surface(vertices, faces, num, seed)
    normals = face_normals(vertices, faces)[fi]
    if vertex_colors is not None and len(vertex_colors) == len(vertices) and vertex_colors.ndim == 2 and vertex_colors.shape[1] >= 3:
        vc = vertex_colors[:, :3] / 255.0
        colors = vc[faces[fi]].mean(axis=1)
    else:
        colors = np.full((num, 3), 0.5, dtype=np.float32)
    return pts.astype(np.float32), normals.astype(np.float32), colors.astype(np.float32)


def prepare_mesh_data(config, mesh: Dict[str, np.ndarray], device, seed: int = 0):
    """mesh: {'vertices' [V,3], 'faces' [F,3] int, 'vertex_normals' [V,3], optional 'vertex_colors' [V,3|4] uint8}.
    Returns (input_data, normalised float64 vertices, faces) -- the reference returns (input_data, mesh, faces)."""
    tr = config.get("training", {}) if isinstance(config, dict) else getattr(config, "training", {})
    num = (tr.get("num_shape_samples", 16384) if isinstance(tr, dict) else getattr(tr, "num_shape_samples", 16384))
    raw = np.asarray(mesh["vertices"])
    faces = np.asarray(mesh["faces"]).astype(np.int64)
    vertex_normals = np.asarray(mesh["vertex_normals"]).astype(np.float32)
    vertices, center, scale = normalize_vertices(raw)                       # float32 chain (ref_pcd)
    mesh_vertices = (raw.astype(np.float64) - center) / scale               # the mesh itself, float64 chain (samples)
    xyz, nrm, rgb = sample_pointcloud_with_albedo(mesh_vertices, faces, num, mesh.get("vertex_colors"), seed)
    dev = torch.device(device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None].float().to(dev)
    samples_dev = torch.from_numpy(xyz).to(dev)
    verts_dev = torch.from_numpy(vertices).to(dev)
    nearest = ops.nearest_point(verts_dev, samples_dev)                     # scripts/inference_with_video_mesh.py:112-115
    rgb_dev = torch.from_numpy(rgb).to(dev)
    input_data = {
        "ref_shape_pcd": t(xyz), "ref_shape_normals": t(nrm), "ref_shape_rgbs": t(rgb),
        "ref_pcd": verts_dev[None].float(), "ref_normal": t(vertex_normals), "ref_rgb": rgb_dev[nearest][None].float(),
        "faces": torch.from_numpy(faces)[None].long().to(dev),
    }
    return input_data, mesh_vertices, faces
