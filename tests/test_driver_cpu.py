"""Host logic around the path: window plan vs the reference driver's golden index maps, work partition,
and the multi-process (gloo, world_size 2) sharding of windows with the final all-gather."""
import json
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN


def _golden():
    return json.load(open(os.path.join(GOLDEN, "chunks.json")))


def _simulate(T, C):
    """What the plan yields when frame t of the video carries the value t (cf. make_chunk_golden.py)."""
    from motion324_amd.inference import plan_windows
    windows, out_map = plan_windows(T, C)
    return [-1 if s is None else windows[s[0]][s[1]] for s in out_map], windows


def test_plan_matches_reference_driver_goldens():
    gold = _golden()
    assert len(gold) >= 50
    for key, expect in gold.items():
        T, C = map(int, key.split(","))
        got, windows = _simulate(T, C)
        assert got == expect, (key, got, expect)
        assert all(len(w) == (C if T > C else T) for w in windows)


def test_known_answers_from_reading_the_reference():
    from motion324_amd.inference import window_starts
    assert window_starts(30, 12) == [0, 11, 18]          # T=30, C=12
    assert window_starts(23, 12) == [0, 11]
    assert window_starts(24, 12) == [0, 11, 12]
    got, windows = _simulate(30, 12)
    assert windows[1] == [0] + list(range(12, 23)) and windows[2] == [0] + list(range(19, 30))
    assert got == [-1] + list(range(1, 30))               # frame 0 := ref_pcd, every other frame exactly once
    assert _simulate(8, 12)[0] == list(range(8))          # single forward keeps the predicted frame 0


def test_merge_is_one_gather_over_the_plan():
    """inference.merge_windows (round 6: one index_select over the (window, slot) pairs, index cached per plan) against the plan read
    slot by slot, for every golden (T, C) pair; the frames the reference overwrites come from ref_pcd."""
    from motion324_amd.inference import merge_windows, plan_windows
    ref = torch.arange(5 * 3, dtype=torch.float32).reshape(1, 5, 3) - 100.0
    for key in _golden():
        T, C = map(int, key.split(","))
        windows, out_map = plan_windows(T, C)
        nW, Cw = len(windows), len(windows[0])
        outs = (torch.arange(nW * Cw, dtype=torch.float32).reshape(nW, Cw, 1, 1) + torch.zeros((1, 1, 5, 3))).contiguous()
        for _ in range(2):                                    # second pass: the cached index
            merged = merge_windows(outs, out_map, ref)
            assert merged.shape == (1, len(out_map), 5, 3)
            for t, slot in enumerate(out_map):
                want = ref[0] if slot is None else outs[slot[0], slot[1]]
                assert torch.equal(merged[0, t], want), (key, t)


def test_graph_shape_key_tells_shapes_byte_frames_and_flags_apart():
    from motion324_amd.graph import shape_key
    base = {"rgb_video": torch.zeros((1, 4, 8, 8, 3)), "ref_pcd": torch.zeros((1, 16, 3))}
    k0 = shape_key(base)
    assert k0 == shape_key({k: v.clone() for k, v in base.items()})
    assert shape_key(dict(base, rgb_video=torch.zeros((1, 4, 8, 8, 3), dtype=torch.uint8))) != k0
    assert shape_key(dict(base, ref_pcd=torch.zeros((1, 17, 3)))) != k0
    assert shape_key(dict(base, m324_keep_reuse=True)) != k0
    assert shape_key(dict(base, m324_mesh_tokens=torch.zeros((1, 64, 8)))) != k0


def test_partition_is_balanced_and_complete():
    from motion324_amd.parallel import counts, partition
    for n in (0, 1, 7, 8, 9, 23, 256):
        for world in (1, 2, 3, 8):
            parts = [list(partition(n, world, r)) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
            assert counts(n, world) == [len(p) for p in parts]
    with pytest.raises(ValueError):
        partition(4, 2, 2)


class _StubModel:
    """Deterministic stand-in for the forward: output frame t = ref_pcd + mean of input frame t."""

    def __call__(self, sample):
        v = sample["rgb_video"]
        out = sample["ref_pcd"][:, None] + v.mean(dim=(2, 3, 4))[:, :, None, None]
        return {"pcd_moved": out}


def _video(T):
    return torch.arange(T, dtype=torch.float32).view(T, 1, 1, 1).expand(T, 4, 4, 3).contiguous() / 10


def _worker(rank, world, port, T, C, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from motion324_amd.inference import run_model_inference
        inp = {"ref_pcd": torch.linspace(-1, 1, 15).view(1, 5, 3)}
        cfg = {"training": {"frames": C, "use_amp": False}}
        out = run_model_inference(_StubModel(), inp, _video(T), cfg, "cpu")
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("T,C", [(30, 12), (77, 8), (9, 12)])
def test_window_sharding_world2_gloo_equals_single_process(T, C):
    from motion324_amd.inference import run_model_inference
    inp = {"ref_pcd": torch.linspace(-1, 1, 15).view(1, 5, 3)}
    cfg = {"training": {"frames": C, "use_amp": False}}
    single = run_model_inference(_StubModel(), inp, _video(T), cfg, "cpu")
    assert single.shape == (1, T, 5, 3)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() + T) % 2000
    mp.spawn(_worker, args=(2, port, T, C, ret), nprocs=2, join=True)
    assert torch.equal(ret[0], single) and torch.equal(ret[1], single)


def test_all_gather_items_uneven_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + os.getpid() % 2000
    mp.spawn(_gather_worker, args=(2, port, ret), nprocs=2, join=True)
    expect = torch.arange(5 * 3, dtype=torch.float32).view(5, 3)
    assert torch.equal(ret[0], expect) and torch.equal(ret[1], expect)


def _gather_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from motion324_amd.parallel import all_gather_items, partition
        full = torch.arange(5 * 3, dtype=torch.float32).view(5, 3)
        mine = partition(5, world, rank)
        ret[rank] = all_gather_items(full[mine.start:mine.stop].clone(), 5)
    finally:
        dist.destroy_process_group()
