"""Pre-step of the path: the repo's restatement of prepare_mesh_data / normalize_mesh /
sample_pointcloud_with_albedo vs a golden produced by the reference's own functions on a stand-in mesh
(tests/golden/make_prestep_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def _gold():
    return dict(np.load(os.path.join(GOLDEN, "prestep.npz")))


def test_normalisation_matches_reference_normalize_mesh():
    from motion324_amd.preprocess import normalize_vertices
    g = _gold()
    v, center, scale = normalize_vertices(g["in_vertices"])
    assert v.dtype == np.float32
    assert np.array_equal(v, g["norm_vertices"]) and np.array_equal(center, g["norm_center"]) and scale == g["norm_scale"]
    assert np.array_equal(v, g["out_ref_pcd"][0])                       # prepare_mesh_data normalises the same way
    assert np.abs(v).max() <= 0.5 and np.abs(v).max() > 0.4999           # unit cube, largest extent touches +-0.5
    assert np.abs(v.max(axis=0) + v.min(axis=0)).max() < 1e-6           # centred box


def test_surface_samples_normals_and_colours_match_reference():
    """Same sampler, then the reference's arithmetic: float64-normalised mesh, face normals, mean vertex colour."""
    from motion324_amd import preprocess as pp
    g = _gold()
    _, center, scale = pp.normalize_vertices(g["in_vertices"])
    mesh_v = (g["in_vertices"].astype(np.float64) - center) / scale
    assert np.allclose(mesh_v, g["mesh_vertices_out"], rtol=0, atol=1e-15)
    xyz, nrm, rgb = pp.sample_pointcloud_with_albedo(mesh_v, g["in_faces"], int(g["num_shape_samples"]), g["in_vertex_colors"],
                                                     seed=int(g["seed"]))
    assert xyz.dtype == nrm.dtype == rgb.dtype == np.float32
    assert np.abs(xyz - g["out_ref_shape_pcd"][0]).max() < 1e-7
    assert np.abs(nrm - g["out_ref_shape_normals"][0]).max() < 1e-6
    assert np.abs(rgb - g["out_ref_shape_rgbs"][0]).max() < 1e-7
    # the sampler itself: uniform over the surface -> the share of samples per face follows the face areas
    pts, fi = pp.sample_surface(mesh_v, g["in_faces"], 200000, seed=5)
    tri = mesh_v[g["in_faces"]]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    share = np.bincount(fi, minlength=len(area)) / len(fi)
    assert np.abs(share - area / area.sum()).max() < 1.5e-3
    # and every sample lies in the plane of its triangle
    n = pp.face_normals(mesh_v, g["in_faces"])[fi]
    assert np.abs(((pts - tri[fi, 0]) * n).sum(axis=1)).max() < 1e-12
    # grey fallback without vertex colours (utils/mesh_processing.py:186-187)
    assert np.all(pp.sample_pointcloud_with_albedo(mesh_v, g["in_faces"], 10)[2] == 0.5)


def test_nearest_sample_colour_oracle_matches_reference_kdtree():
    """numpy brute force (the checker of the HIP kernel) == the reference's cKDTree assignment."""
    g = _gold()
    q, r = g["out_ref_pcd"][0], g["out_ref_shape_pcd"][0]
    d = ((q[:, None, :] - r[None, :, :]) ** 2).sum(-1)
    assert np.array_equal(g["out_ref_shape_rgbs"][0][d.argmin(axis=1)], g["out_ref_rgb"][0])


@pytest.mark.gpu
def test_prepare_mesh_data_on_the_device_matches_reference():
    from motion324_amd.preprocess import prepare_mesh_data
    g = _gold()
    mesh = {"vertices": g["in_vertices"], "faces": g["in_faces"], "vertex_normals": g["in_vertex_normals"],
            "vertex_colors": g["in_vertex_colors"]}
    inp, mesh_v, faces = prepare_mesh_data({"training": {"num_shape_samples": int(g["num_shape_samples"])}}, mesh, "cuda",
                                           seed=int(g["seed"]))
    assert set(inp) == {"ref_shape_pcd", "ref_shape_normals", "ref_shape_rgbs", "ref_pcd", "ref_normal", "ref_rgb", "faces"}
    for k, t in inp.items():
        want = g["out_" + k]
        assert t.is_cuda and tuple(t.shape) == want.shape and str(t.dtype).endswith(str(want.dtype)), k
        if k == "faces":
            assert np.array_equal(t.cpu().numpy(), want)
        else:
            assert np.abs(t.cpu().numpy() - want).max() < 1e-6, k
    assert np.array_equal(inp["ref_rgb"].cpu().numpy(), g["out_ref_rgb"])         # the nearest-sample kernel picks the same samples


@pytest.mark.gpu
def test_nearest_point_kernel_large():
    from motion324_amd import ops
    torch.manual_seed(0)
    q, r = torch.rand(5000, 3), torch.rand(16384, 3)
    want = torch.cdist(q.double(), r.double()).argmin(dim=1)
    got = ops.nearest_point(q.cuda(), r.cuda()).cpu()
    same = got == want
    # fp32 distances may swap two near-equidistant samples: the chosen point must then be (almost) as close
    dq = (q - r[got]).norm(dim=1) - (q - r[want]).norm(dim=1)
    assert same.float().mean() > 0.999 and float(dq.abs().max()) < 1e-6
