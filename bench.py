#!/usr/bin/env python3
"""Headline benchmark: frames/sec of Motion_Latent_Model.forward on the BASELINE.json clip
(32 frames x 2048 mesh points x 512x512 video, 4096 surface samples, batch 1, bf16) on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode infer|train|frame-parallel]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run ...`, launched before this process touches the GPU) and returns its exit code.

Modes (BASELINE.json `configs`):
  infer (default, configs[1], c2): one "step" = one forward of the hot path over one 32-frame clip per GPU, inputs resident
      in HBM, hipGraph replay.  Clips are independent: no collective inside the forward, the per-clip
      [T,N,3] offsets are all-gathered at the end of each step.  Weak scaling.  THE headline line.
  train (configs[2] / [3], c3 / c4): one step = forward + hand-written backward + bucketed gradient all-reduce (overlapped
      with the backward) + fused AdamW on dyscene.yaml shapes (12 frames, 4096 points, 224x224), --batch per GPU (8 / 32).
  frame-parallel (configs[4], c5): one 256-frame clip, frames sharded over the ranks with exact single-GPU semantics
      (K/V all-gather per global block).  Strong scaling; at N = 1 the whole clip runs on one GPU.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
PROFILE_ROUND = "r06"          # profiles/<round>_* hold the rocprofv3 summaries the roofline rows are checked against

WORKLOAD = dict(B=1, T=32, N=2048, S=4096, HW=512, frames=32)   # BASELINE.json configs[1]


def algorithmic_flops(B, T, N, S, d=768, K=64, g=16, n_layer=16, pcd_layers=4, dino_depth=12):
    """Forward FLOPs (2*MAC) of the reference's arithmetic, SURVEY.md 8(d) (incl. its per-frame recompute
    of the decoder point features): 7.527 TFLOP for the c2 clip, 208.0 for the 256-frame clip."""
    lin = lambda m, i, o: 2.0 * m * i * o
    att = lambda b, q, k: 4.0 * b * (d // 64) * q * k * 64
    blk = lambda m: lin(m, d, 3 * d) + lin(m, d, d) + 2 * lin(m, d, 4 * d)
    L = 4 + K + g * g
    f = B * (lin(S, 51, d) + lin(S, d + 6, d))
    f += B * (2 * lin(K, d, d) + 2 * lin(S, d, d) + att(1, K, S) + 2 * lin(K, d, 4 * d))
    f += pcd_layers * B * (blk(K) + att(1, K, K))
    f += B * T * (lin(g * g, 588, d) + dino_depth * (blk(g * g + 1) + att(1, g * g + 1, g * g + 1)))
    f += (n_layer // 2) * B * (blk(T * L) + att(1, T * L, T * L))
    f += (n_layer // 2) * B * (blk(T * L) + T * att(1, L, L))
    f += B * T * (2 * lin(N, d, d) + 2 * lin(K, d, d) + 2 * lin(N, d, 4 * d) + att(1, N, K))
    f += B * T * (lin(N, d, d) + lin(N, d, 3))
    f += B * T * (lin(N, 51, d) + lin(N, d + 6, d))
    return f


def build_model(device, frames, train=False):
    import motion324_amd as m
    from motion324_amd import synth
    cfg = synth.make_config(frames=frames)
    model = m.Motion_Latent_Model(cfg)
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.Dims(frames=frames), seed=0).items()}
    model.load_state_dict(sd, strict=False)
    model = (model.train() if train else model.eval()).to(device)
    model.auto_graph = False              # the bench drives its graphs and its eager (instrumented) passes explicitly
    return model, sd


# ----------------------------------------------------------------------------------------------- CPU baseline
def _cpu_identity():
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(cores) or logical), logical


def cpu_baseline(sd, sample_np, frames, budget_s=45.0):
    """The CPU oracle (fp32 restatement of the reference path, BASELINE.md section 3 protocol) timed on this host's cores:
    thread count chosen by a measured sweep on a 4-frame sub-clip (torch's CPU kernels stop scaling, then collapse, far
    below a 256-thread host -- the sweep is reported), then 1 warm-up (the sub-clip) + median of up to 3 forwards of
    the full 32-frame clip, bounded by `budget_s` of CPU work."""
    from oracle import ref_forward as oracle
    cpu_model, physical, logical = _cpu_identity()
    sample = oracle.to_torch(sample_np)
    T = sample["rgb_video"].shape[1]
    sub = dict(sample)
    sub["rgb_video"] = sample["rgb_video"][:, :4]
    cands = sorted({c for c in (8, 16, 32, 64, physical) if 1 <= c <= logical})
    sweep = {}
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            oracle.forward(sd, sub, frames=frames)                        # warm-up at this thread count
            t0 = time.perf_counter()
            oracle.forward(sd, sub, frames=frames)
            sweep[c] = time.perf_counter() - t0
            if sweep[c] > 3.0 * min(sweep.values()):
                break                                                       # collapsing: stop climbing
        threads = min(sweep, key=sweep.get)
        torch.set_num_threads(threads)
        times, out = [], None
        spent = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            out = oracle.forward(sd, sample, frames=frames)["pcd_moved"]
            dt = time.perf_counter() - t0
            times.append(dt)
            spent += dt
            if spent + dt > budget_s:
                break
    med = sorted(times)[len(times) // 2]
    sweep_s = ", ".join(f"{c}t {v * 1e3:.0f} ms" for c, v in sweep.items())
    return {"value": round(T / med, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model, "physical_cores": physical, "logical_cpus": logical,
            "sample": f"median of {len(times)} forwards of the full {T}-frame clip after a warm-up (fp32 oracle, torch CPU ops, "
                      f"{threads} threads of {physical} physical cores; 4-frame thread sweep: {sweep_s}); {med:.1f} s per clip"}, out


# ----------------------------------------------------------------------------------------------- launch plumbing
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_spawn(args) -> None:
    """--gpus N > 1 without a torchrun environment: start the N ranks as a CHILD process tree (never an exec: this
    process may not have touched the GPU yet, and stays that way) and exit with the child's code."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


class Dist:
    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={self.world} ranks")
        # M324_BENCH_BACKEND=gloo runs the N > 1 control flow with several ranks on ONE device (rehearsal of the launch
        # line on a single-GPU box; the numbers mean nothing): the driver's real runs use RCCL ("nccl"), one rank per GPU
        self.backend = os.environ.get("M324_BENCH_BACKEND", "nccl")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU path to benchmark")
        if self.backend != "nccl":
            self.local %= torch.cuda.device_count()
        elif self.world > torch.cuda.device_count():
            raise SystemExit(f"bench.py: {self.world} ranks but only {torch.cuda.device_count()} GPUs are visible")
        self.dev = torch.device("cuda", self.local)
        torch.cuda.set_device(self.dev)
        self.comm_ranks = 1
        # M324_BENCH_COLLECT=1 on ONE rank: a 1-rank RCCL communicator with every collective forced (identities there): the
        # N > 1 code path -- segmented graph chain, sharded video, side-stream exchanges -- rehearsed on a single-GPU box
        self.force_collect = self.world == 1 and os.environ.get("M324_BENCH_COLLECT") == "1"
        if self.force_collect:
            import torch.distributed as dist
            from motion324_amd import parallel
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            dist.init_process_group(self.backend, rank=0, world_size=1, **({"device_id": self.dev} if self.backend == "nccl" else {}))
            parallel.ALWAYS_COLLECT = True
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
            # what the communicator itself reports: one all-reduce of ones over RCCL must give N
            ones = torch.ones(1, device=self.dev)
            dist.all_reduce(ones)
            self.comm_ranks = int(ones.item())
            if self.comm_ranks != args.gpus or dist.get_world_size() != args.gpus:
                raise SystemExit(f"bench.py: communicator has {self.comm_ranks} ranks, expected {args.gpus}")

    def fence(self):
        torch.cuda.synchronize()
        if self.world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(self, dt: float) -> float:
        if self.world == 1:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.world > 1 or self.force_collect:
            torch.distributed.destroy_process_group()


def timed(dist_: Dist, step, steps: int):
    """EXACTLY `steps` calls of step() between two fences; returns (wall seconds max over ranks, host enqueue seconds)."""
    dist_.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_enq = time.perf_counter() - t0
    dist_.fence()
    return dist_.max_over_ranks(time.perf_counter() - t0), t_enq


def sustained(dist_: Dist, step, seconds: float):
    """>= `seconds` of back-to-back steps (clock / power throttling shows up here, not in a 0.2 s timed region)."""
    if seconds <= 0:
        return None
    dist_.fence()
    t0 = time.perf_counter()
    n = 0
    while True:
        for _ in range(32):
            step()
        n += 32
        torch.cuda.synchronize()              # bound the host's run-ahead so that the loop ends near `seconds`
        stop = time.perf_counter() - t0 >= seconds
        if dist_.world > 1:                   # every rank must leave after the same step (the step holds a collective)
            flag = torch.tensor([1.0 if stop else 0.0], device=dist_.dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            stop = float(flag.item()) > 0
        if stop:
            break
    dist_.fence()
    dt = dist_.max_over_ranks(time.perf_counter() - t0)
    return n, dt


# ----------------------------------------------------------------------------------------------- roofline rows
def roofline_rows(rec, steps: int):
    """Per-symbol rows from the HIP-event recorder: (kernel template + grid) -> launches, avg us, algorithmic FLOP / bytes
    per launch, fraction of the dense peak.  The same symbols appear in profiles/<round>_*_kernel_stats.md."""
    rows = {}
    for (cls, tag), v in rec.by_tag().items():
        sym = tag.split(" | ")[0] if " | " in tag else cls
        shape = tag.split(" | ")[1] if " | " in tag else tag
        r = rows.setdefault((cls, sym), {"class": cls, "symbol": sym, "launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0,
                                         "shapes": set()})
        r["launches"] += v["launches"]
        r["total_ms"] += v["total_ms"]
        r["flops"] += v["flops"]
        r["bytes"] += v.get("bytes", 0.0)
        r["shapes"].add(shape.split(" bias")[0].split(" gelu")[0].split(" res")[0].split(" out=")[0].split(" qkv")[0])
    out = []
    for r in rows.values():
        peak = PEAK_F32_TFLOPS if r["class"].endswith("f32") else PEAK_BF16_TFLOPS
        tf = r["flops"] / max(r["total_ms"], 1e-9) / 1e9
        gbs = r["bytes"] / max(r["total_ms"], 1e-9) / 1e6
        # which roofline bounds the launch: algorithmic intensity against the ridge peak FLOP/s : 8 TB/s (312 FLOP/B in bf16)
        ai = r["flops"] / max(r["bytes"], 1.0)
        hbm = ai < peak * 1e12 / (PEAK_HBM_GBS * 1e9)
        row = {"symbol": r["symbol"], "shapes": sorted(r["shapes"]), "launches_per_step": r["launches"] // steps,
               "avg_us": round(r["total_ms"] / r["launches"] * 1e3, 2), "ms_per_step": round(r["total_ms"] / steps, 3),
               "gflop_per_launch": round(r["flops"] / r["launches"] / 1e9, 2),
               "mbytes_per_launch": round(r["bytes"] / r["launches"] / 1e6, 2),
               "bound": "hbm" if hbm else "mfma", "tflops": round(tf, 1), "gbytes_per_s": round(gbs, 1),
               "frac": round(gbs / PEAK_HBM_GBS, 4) if hbm else round(tf / peak, 4)}
        if hbm and r["flops"] > 0:
            row["frac_of_mfma_peak"] = round(tf / peak, 4)
        out.append(row)
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


def class_totals(summ, steps: int, peak: float):
    out = {}
    for k, v in summ.items():
        t = {"ms_per_step": round(v["total_ms"] / steps, 3)}
        if v["flops"] > 0:
            t["tflops"] = round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 1)
            t["frac"] = round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12 / peak, 4)
        else:                                   # memory-bound passes (LayerNorm, statistics merges): against 8 TB/s
            t["gbytes_per_s"] = round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1)
            t["frac_of_hbm_peak"] = round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
        out[k] = t
    return out


def committed_traffic(symbol: str):
    """HBM bytes per launch of `symbol` from the committed rocprofv3 PMC passes (counters cannot be read in-process):
    profiles/<round>_traffic.json, written by tools/pmc_traffic.py from FETCH_SIZE / WRITE_SIZE passes."""
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03", "r02", "r01"):
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", f"{rnd}_traffic.json")))
        except (OSError, ValueError):
            continue
        key = symbol.split(" grid=")[0]
        for k, v in tj.items():
            if k == symbol or k == key or k.split(" grid=")[0] == key:
                return v.get("traffic_bytes_per_launch"), f"profiles/{rnd}_traffic.json"
    return None, None


# ----------------------------------------------------------------------------------------------- secondary measurements
def _event_time_ms(fn, steps: int) -> float:
    """`steps` calls of fn() between two HIP events on the current stream (ms per call)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def decoder_block_replay(model, sample, steps: int, q_mode: str = "pair"):
    """The north-star block (decoder cross-attention: k|v and q projections, attention, out-projection, MLP -- reference
    transformer.py:365-377 via Pcd_motion.py:556-561) as the product runs it: its launches captured into a hipGraph of their
    own and replayed back to back between two HIP events.  The block reads the trunk's REAL output stream of this clip
    (taken from one eager forward).  (The eager per-launch events of the roofline pass also bracket this block, but those
    carry the host's launch gaps and an event pair around every kernel.)"""
    from motion324_amd.prepared import Prepared, compute_dtype
    dev = sample["ref_pcd"].device
    P = Prepared.for_module(model, dev, compute_dtype())
    cap = {}
    model._capture = cap
    try:
        with torch.no_grad():
            model(sample)
    finally:
        model._capture = None
    C, K = model.embed_dim, model.num_learnable_tokens
    B, T, Hh, Ww, _ = sample["rgb_video"].shape
    N = sample["ref_pcd"].shape[1]
    Lt = 4 + K + model.num_patches_h * model.num_patches_w
    tok = cap["trunk_out"].reshape(-1, C).contiguous()
    dec = model.decoder_cross_attn
    with torch.no_grad():
        pf = model._point_features(P, sample["ref_pcd"][0].float().contiguous(), sample["ref_normal"][0].float().contiguous(),
                                   sample["ref_rgb"][0].float().contiguous())

        branch = torch.cuda.Stream()

        def block():
            # q_mode "pair" (default): the block as the product's forward runs it when the q projection is inside the block (every
            # eager forward; under graph capture Pcd_motion._forward hoists it onto the shape-encoder branch instead, where it costs
            # the block nothing): norm_q + norm_kv in one launch, the q + k|v projections in one launch (transformer.project_q_kv).
            # "branch": round 3's stand-in for the hoist -- the q side on a second graph branch beside the k|v side;
            # "serial": four launches on one stream.
            main = torch.cuda.current_stream()
            if q_mode == "pair":
                Q, Kd, Vd = dec.project_q_kv(P, pf, N, tok, B * T, K, row_map=(K, Lt, 4))
                return model.decoder_block(P, Kd[:T], Vd[:T], pf, Q)
            if q_mode == "serial":
                Q = dec.project_q(P, pf, 1, N)
                Kd, Vd = dec.project_kv(P, tok, B * T, K, row_map=(K, Lt, 4))
                return model.decoder_block(P, Kd[:T], Vd[:T], pf, Q)
            branch.wait_stream(main)
            with torch.cuda.stream(branch):
                Q = dec.project_q(P, pf, 1, N)
            Kd, Vd = dec.project_kv(P, tok, B * T, K, row_map=(K, Lt, 4))
            main.wait_stream(branch)
            return model.decoder_block(P, Kd[:T], Vd[:T], pf, Q)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                block()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = block()
        for _ in range(100):                                 # ~80 ms: the clock has settled before the first timed round (three replays
            g.replay()                                       # did not: the first of five rounds used to be the slow one, by up to 10 %)
        in_order = [_event_time_ms(g.replay, max(steps, 20)) for _ in range(5)]
        rounds = sorted(in_order)
        ms = rounds[2]                                       # median of five rounds of >= 20 replays
    del keep
    flops = model.decoder_block_flops(1, T, N)
    fexe = model.decoder_block_flops_executed(1, T, N)
    return {"ms_per_step": round(ms, 4), "algorithmic_gflop": round(flops / 1e9, 1), "tflops": round(flops / ms / 1e9, 1),
            "frac_of_bf16_peak": round(flops / ms / 1e9 / PEAK_BF16_TFLOPS, 4),
            "executed_gflop": round(fexe / 1e9, 1), "frac_executed_flops": round(fexe / ms / 1e9 / PEAK_BF16_TFLOPS, 4),
            "rounds_ms": [round(r, 4) for r in rounds], "rounds_in_order_ms": [round(r, 4) for r in in_order],
            "q_mode": q_mode,
            "timing": "hipGraph of the block alone (norm_q + norm_kv, q + k|v projections -- one launch each, as in the product's forward "
                      "when the q projection is not hoisted --, attention, out-projection, MLP), replayed back to back between two HIP events (median of five rounds); input = this "
                      "clip's trunk output"}


def secondary_measurements(args, D, model, sd, sample, sample_np, ref_out, headline_ms=None):
    """BASELINE's other configurations and the headline's precision variants, driver-visible in the default command
    (each a few steps; a failure is reported in place, never raised): the headline without its two narrower-than-reference
    shortcuts, fp32 parity mode on c2, the c3 training step, the 256-frame clip on one GPU, and the PCIe hand-over."""
    import gc
    import motion324_amd as m
    import motion324_amd.Pcd_motion as pm
    from motion324_amd import synth
    w = WORKLOAD
    dev = D.dev
    out = {}

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:                      # noqa: BLE001 -- a secondary number must never cost the headline line
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
        gc.collect()
        torch.cuda.empty_cache()

    def rel(o):
        return None if ref_out is None else round(float((o.double().cpu() - ref_out.double()).norm() / ref_out.double().norm()), 6)

    def clip_rate(steps):
        fast = m.GraphedForward(model)
        with torch.no_grad():
            clip = fast.static_inputs(sample)
            for _ in range(2):
                fast(clip)
            ms = _event_time_ms(lambda: fast(clip), steps)
            o = fast(clip).pcd_moved.clone()
        return ms, o

    def strict():
        """the headline with the decoder's residual stream in fp32 and the head's intermediate stored (M324_BF16_DECODER=0
        M324_FUSE_HEAD=0): the reference's own precision layout under autocast"""
        old = pm.BF16_DECODER_STREAM, pm.FUSE_HEAD_N3
        pm.BF16_DECODER_STREAM, pm.FUSE_HEAD_N3 = False, False
        try:
            ms, o = clip_rate(10)
        finally:
            pm.BF16_DECODER_STREAM, pm.FUSE_HEAD_N3 = old
        return {"value": round(w["B"] * w["T"] / ms * 1e3, 2), "unit": "frames/s", "ms_per_step": round(ms, 3),
                "rel_err_vs_cpu_oracle": rel(o), "switches": {"M324_BF16_DECODER": 0, "M324_FUSE_HEAD": 0}}

    def fp32():
        """fp32 parity mode (v_mfma_f32_32x32x2_f32 everywhere): the mode the 1e-3 gate of the north-star is met in"""
        m.set_precision("fp32")
        try:
            ms, o = clip_rate(4)
        finally:
            m.set_precision(args.precision)
        flops = algorithmic_flops(w["B"], w["T"], w["N"], w["S"])
        return {"value": round(w["B"] * w["T"] / ms * 1e3, 2), "unit": "frames/s", "ms_per_step": round(ms, 2), "dtype": "f32",
                "rel_err_vs_cpu_oracle": rel(o), "end_to_end_tflops": round(flops / ms / 1e9, 1),
                "frac_of_f32_peak": round(flops / ms / 1e9 / PEAK_F32_TFLOPS, 4)}

    def h2d():
        host = torch.from_numpy(sample_np["rgb_video"]).pin_memory()
        dst = torch.empty_like(sample["rgb_video"])
        dst.copy_(host, non_blocking=True)
        ms = _event_time_ms(lambda: dst.copy_(host, non_blocking=True), 5)
        return {"h2d_ms": round(ms, 3), "mbytes": round(host.numel() * 4 / 1e6, 1), "gbytes_per_s": round(host.numel() * 4 / ms / 1e6, 1),
                "note": "pinned host -> HBM copy of the clip's fp32 frames; never part of `value`"}

    def train_c3(Bt=8, timed_steps=5):
        """BASELINE configs[2]: dyscene.yaml shapes, batch_size_per_gpu = 8, forward + backward + fused AdamW
        (Bt = 32: configs[3]'s per-GPU workload on this one GPU -- no gradient exchange, that is the 8-GPU run's part)"""
        from motion324_amd import training
        from motion324_amd.optim import FusedAdamW, backward_completion_order, cosine_with_warmup
        T, N, HW = 12, 4096, 224
        tm, _ = build_model(dev, T, train=True)
        sn = synth.synth_inputs(Bt, T, N, N, HW, seed=1, with_target=True)
        smp = {k: torch.from_numpy(v).to(dev) for k, v in sn.items()}
        opt = FusedAdamW(tm.named_parameters(), lr=4e-4, betas=(0.9, 0.95), weight_decay=0.05, grad_clip_norm=1.0,
                         allowed_gradnorm_factor=5.0, order=backward_completion_order(tm))
        it = [0]
        losses = []

        def step():
            loss, _, G = training.forward_backward(tm, smp, sink=opt)
            opt.finish_reduce()
            opt.step(lr=cosine_with_warmup(it[0], 1000, 30000, 4e-4))
            it[0] += 1
            losses.append(loss)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(timed_steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / timed_steps * 1e3
        fwd = algorithmic_flops(Bt, T, N, N)
        # executed FLOPs of a step = forward + 2 x backward of the trainable part (its internals are kept, not recomputed:
        # motion324_amd/training.py, M324_TRAIN_STORE; with =0 a fourth, recompute pass runs) + the frozen DINO forward
        dino = algorithmic_flops(Bt, T, N, N) - algorithmic_flops(Bt, T, N, N, dino_depth=0)
        kept = training.TRAIN_STORE != "0"
        step_flops = dino + (3.0 if kept else 4.0) * (fwd - dino)
        res = {"value": round(Bt / ms * 1e3, 2), "unit": "samples/s", "ms_per_step": round(ms, 2), "batch_size_per_gpu": Bt,
               "forward_tflop_per_step": round(fwd / 1e12, 2), "tflops": round(step_flops / ms / 1e9, 1),
               "frac_of_bf16_peak": round(step_flops / ms / 1e9 / PEAK_BF16_TFLOPS, 4),
               "flop_model": ("DINO forward + 3 x trainable forward (forward with kept internals, 2 x backward)" if kept else
                              "DINO forward + 4 x trainable forward (forward, recompute, 2 x backward)"),
               "activations": "kept in HBM while they fit half of the free memory" if kept else "checkpoint per block + recompute",
               "loss": round(float(losses[-1]), 6), "finite": bool(all(math.isfinite(float(l)) for l in losses)),
               "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
        kb = getattr(training, "LAST_STEP_BLOCKS", None)
        if kb:
            res["blocks_kept_vs_recomputed"] = kb
        del tm, opt, smp
        return res

    def train_c4():
        """BASELINE configs[3]'s per-GPU workload (dyscene.yaml, batch_size_per_gpu = 32) on ONE GPU, three timed steps"""
        torch.cuda.reset_peak_memory_stats()
        r = train_c3(32, 3)
        r["note"] = "one GPU, no gradient exchange: the 8-GPU all-reduce of configs[3] is bench.py --mode train --gpus 8"
        return r

    def clip256():
        """BASELINE configs[4] on ONE GPU: the 256-frame clip (82 944 trunk tokens) through the same forward"""
        T = 256
        cm, _ = build_model(dev, T)
        sn = synth.synth_inputs(1, T, 2048, 4096, 512, seed=1)
        smp = {k: torch.from_numpy(v).to(dev) for k, v in sn.items()}
        fast = m.GraphedForward(cm, warmup=1)
        with torch.no_grad():
            clip = fast.static_inputs(smp)
            fast(clip)
            ms = _event_time_ms(lambda: fast(clip), 3)
            fin = bool(torch.isfinite(fast(clip).pcd_moved).all())
        flops = algorithmic_flops(1, T, 2048, 4096)
        res = {"value": round(T / ms * 1e3, 2), "unit": "frames/s", "ms_per_step": round(ms, 2), "frames": T,
               "end_to_end_tflops": round(flops / ms / 1e9, 1), "frac_of_bf16_peak": round(flops / ms / 1e9 / PEAK_BF16_TFLOPS, 4),
               "finite": fin}
        del cm, fast, smp
        return res

    def resized():
        """the released checkpoint's setting (scripts/4D_from_existing.sh: a model trained with training.frames = 12 run on a
        32-frame clip): the 3-D position table is resized 12 -> 32 frames (Pcd_motion.py:221-228), cached per clip length"""
        rm, _ = build_model(dev, 12)
        fast = m.GraphedForward(rm, warmup=1)
        with torch.no_grad():
            clip = fast.static_inputs(sample)
            fast(clip)
            ms = _event_time_ms(lambda: fast(clip), 10)
            fin = bool(torch.isfinite(fast(clip).pcd_moved).all())
        del rm, fast
        return {"value": round(w["B"] * w["T"] / ms * 1e3, 2), "unit": "frames/s", "ms_per_step": round(ms, 3), "finite": fin,
                "note": "training.frames = 12, clip of 32 frames (parity of the resize: goldens tiny_resize and c1)"}

    def long_video():
        """SURVEY 8(f) row 1, the caller north_star names: motion324_amd.inference.run_model_inference (the reference's
        scripts/inference_with_video_mesh.py:132-256) on a 256-frame HOST-resident video cut into 32-frame windows -- wall clock
        around the whole call (window plan, uploads, forwards, merge), i.e. what a user of the script sees.  Nine windows of 32
        frames produce the 256 output frames (every window re-runs the anchor frame, the last one overlaps its neighbour)."""
        from motion324_amd.inference import plan_windows, run_model_inference
        T, C = 256, w["T"]
        g = torch.Generator().manual_seed(11)
        vid8 = torch.randint(0, 256, (T, w["HW"], w["HW"], 3), generator=g, dtype=torch.uint8).pin_memory()
        vidf = (vid8.float() / 255.0).pin_memory()
        inp = {k: v for k, v in sample.items() if k != "rgb_video"}
        cfg = {"training": {"frames": C, "use_amp": args.precision == "bf16"}}
        n_win = len(plan_windows(T, C)[0])

        def wall(video, reps=3, **kw):
            model._drop_auto_graph()                    # every row starts from a model without captured graphs
            for _ in range(3):                          # both window shapes reach their hipGraph (third call of a shape on)
                o = run_model_inference(model, inp, video, cfg, dev, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                o = run_model_inference(model, inp, video, cfg, dev, **kw)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3, o
        rows = {}
        was = model.auto_graph
        model.auto_graph = True                         # the scripts' plain `model(sample)` loop: graph replay from the third call on
        try:
            ms_u8, o_u8 = wall(vid8)
            ms_f, o_f = wall(vidf)
            ms_plain, o_p = wall(vidf, pipelined=False)
            ms_pageable, _ = wall(vidf.clone(), reps=2)     # a caller that did not pin its frames: through the driver's bounce buffers
        finally:
            model.auto_graph = was
            model._drop_auto_graph()
        for name, ms in (("pipelined_uint8_frames", ms_u8), ("pipelined_fp32_frames", ms_f), ("plain_loop_fp32_frames", ms_plain),
                         ("pipelined_fp32_frames_pageable_host", ms_pageable)):
            rows[name] = {"ms_per_video": round(ms, 2), "video_frames_per_s": round(T / ms * 1e3, 1),
                          "forwarded_frames_per_s": round(n_win * C / ms * 1e3, 1), "ms_per_window": round(ms / n_win, 3)}
        res = {"value": rows["pipelined_uint8_frames"]["forwarded_frames_per_s"], "unit": "frames/s (forwarded; H2D, host work and merge included)",
               "frames": T, "window": C, "windows": n_win, "rows": rows,
               "identical_results": bool(torch.equal(o_u8, o_f) and torch.equal(o_f, o_p)), "finite": bool(torch.isfinite(o_u8).all()),
               "note": "wall clock around run_model_inference on a pinned host video; plain_loop = the reference's literal loop (one "
                       "synchronous upload per window, shape encoder and anchor frame recomputed per window); the forwards of all rows "
                       "are hipGraph replays (the model's automatic graph, third call of a shape on)"}
        if headline_ms:
            res["headline_ms_per_clip"] = round(headline_ms, 3)
            res["forwarded_rate_vs_headline"] = round(rows["pipelined_uint8_frames"]["forwarded_frames_per_s"] / (w["B"] * w["T"] / headline_ms * 1e3), 4)
        return res

    guarded("headline_without_precision_shortcuts", strict)
    guarded("long_video_driver", long_video)
    guarded("c2_with_training_frames_12_pos_embed_resized", resized)
    guarded("fp32_parity_mode_c2", fp32)
    guarded("h2d", h2d)
    guarded("train_c3", train_c3)
    guarded("train_c4_per_gpu", train_c4)
    guarded("clip_256_frames_one_gpu", clip256)
    return out


# ----------------------------------------------------------------------------------------------- modes
def roofline_from(rec, steps: int, precision: str, note: str):
    """The `roofline` object of the secondary modes: dominant symbol of one instrumented eager pass (HIP events per launch)."""
    rows = [r for r in roofline_rows(rec, steps) if not r["symbol"].startswith("stage:")]
    if not rows:
        return None
    dom = rows[0]
    peak = PEAK_BF16_TFLOPS if precision == "bf16" else PEAK_F32_TFLOPS
    traffic, traffic_src = committed_traffic(dom["symbol"])
    summ = {k: v for k, v in rec.summary().items() if not k.startswith("stage:")}
    return dict(dominant(dom, peak), traffic=traffic, traffic_source=traffic_src, timing=note, by_symbol=rows[:8],
                class_totals=class_totals(summ, steps, peak))


def dominant(dom, peak):
    """The contract's roofline fields for the dominant symbol."""
    if dom["bound"] == "hbm":
        head = {"bound": "hbm", "kernel": dom["symbol"], "achieved": dom["gbytes_per_s"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": dom["frac"]}
    else:
        head = {"bound": "mfma", "kernel": dom["symbol"], "achieved": dom["tflops"], "peak": peak, "unit": "TFLOP/s",
                "frac": round(dom["tflops"] / peak, 4)}
    head.update({"avg_launch_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"],
                 "algorithmic_gflop_per_launch": dom["gflop_per_launch"], "algorithmic_mbytes_per_launch": dom["mbytes_per_launch"]})
    return head


def run_infer(args, D: Dist):
    import motion324_amd as m
    from motion324_amd import parallel, synth
    from motion324_amd.timing import Recorder
    w = WORKLOAD
    world, rank, dev = D.world, D.rank, D.dev
    model, sd = build_model(dev, w["frames"])
    # every rank works on its own clip: same shape, different seed
    sample_np = synth.synth_inputs(w["B"], w["T"], w["N"], w["S"], w["HW"], seed=1 + rank)
    sample = {k: torch.from_numpy(v).to(dev) for k, v in sample_np.items()}
    m.set_precision(args.precision)
    from motion324_amd import image_encoder
    image_encoder.TWO_STREAMS = args.dino_streams == 2

    fast = None if args.eager else m.GraphedForward(model)
    # --clips-in-flight 2: a second graph with its own static buffers on a second stream; step i runs on stream i % 2
    lanes = [(torch.cuda.current_stream(), fast)]
    if args.clips_in_flight == 2 and fast is not None and world == 1:     # N > 1: one communicator, one stream of collectives
        lanes = [(torch.cuda.Stream(), fast), (torch.cuda.Stream(), m.GraphedForward(model))]
    step_no = [0]
    gathered = [None]
    last = [None]
    bound = {}

    def handed_over(fwd):
        """the clip, written once into the graph's own input buffers (zero-copy handover: a replay then starts without the
        100.7 MB device-to-device copy of the frames that ``fwd(sample)`` would make every step)"""
        if id(fwd) not in bound:
            with torch.no_grad():
                bound[id(fwd)] = fwd.static_inputs(sample)
        return bound[id(fwd)]

    def gather(out):
        if world > 1:
            if gathered[0] is None:
                gathered[0] = torch.empty((world,) + tuple(out.shape), dtype=out.dtype, device=dev)
            parallel.all_gather_into(gathered[0], out.contiguous())

    def step(eager=False):
        stream, fwd = lanes[step_no[0] % len(lanes)]
        step_no[0] += 1
        with torch.no_grad():
            if eager or fast is None:
                out = model(sample).pcd_moved
                gather(out)
            else:
                clip = handed_over(fwd)
                with torch.cuda.stream(stream):
                    out = fwd(clip).pcd_moved
                    gather(out)
        last[0] = out
        return out

    for _ in range(args.warmup):
        step()
    dt, t_enq = timed(D, step, args.steps)
    out = last[0]
    sus = sustained(D, step, args.sustain)
    # serving-throughput variant (reported next to the headline, never instead of it): two clips in flight -- consecutive
    # steps alternate between two streams / graphs, so one clip's kernels fill the partly filled last rounds of the other's
    two = None
    if world == 1 and fast is not None and len(lanes) == 1 and not args.no_two_in_flight:
        single = list(lanes)
        lanes[:] = [(torch.cuda.Stream(), fast), (torch.cuda.Stream(), m.GraphedForward(model))]
        for _ in range(4):
            step()
        dt2, _ = timed(D, step, args.steps)
        two = {"value": round(w["B"] * w["T"] * args.steps / dt2, 2), "unit": "frames/s", "ms_per_step": round(dt2 / args.steps * 1e3, 3),
               "note": "two clips in flight on two HIP streams (throughput mode; a clip's latency does not improve)"}
        torch.cuda.synchronize()
        lanes[:] = single
    blk_replay = None
    if world == 1 and fast is not None:
        try:
            blk_replay = decoder_block_replay(model, sample, args.steps)
            # the same block with the decoder's residual stream in fp32 (M324_BF16_DECODER=0): what the precision shortcut buys
            import motion324_amd.Pcd_motion as pm_
            if pm_.BF16_DECODER_STREAM:
                pm_.BF16_DECODER_STREAM = False
                try:
                    r32 = decoder_block_replay(model, sample, args.steps)
                    blk_replay["fp32_decoder_stream"] = {k: r32[k] for k in ("ms_per_step", "frac_of_bf16_peak", "frac_executed_flops")}
                finally:
                    pm_.BF16_DECODER_STREAM = True
        except Exception as e:                          # noqa: BLE001
            blk_replay = {"error": f"{type(e).__name__}: {e}"[:300]}
    # per-kernel HIP-event timing (roofline object): the same K steps again, launched eagerly on the same
    # stream with an event pair around every GEMM / attention launch (events cannot sit inside a graph)
    rec = Recorder()
    with rec:
        for _ in range(args.steps):
            step(eager=True)
    D.fence()

    line = None
    if rank == 0:
        frames_per_step = world * w["B"] * w["T"]
        ms = dt / args.steps * 1e3
        value = frames_per_step * args.steps / dt
        summ = rec.summary()
        stages = {k: summ.pop(k) for k in list(summ) if k.startswith("stage:")}
        rows = [r for r in roofline_rows(rec, args.steps) if not r["symbol"].startswith("stage:")]
        dom = rows[0]                                  # dominant SYMBOL (kernel template + grid) by time per step
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
        traffic, traffic_src = committed_traffic(dom["symbol"])
        roof = dict(dominant(dom, peak), traffic=traffic, traffic_source=traffic_src,
                    timing="HIP events around every launch of an eager pass on the launch stream; the same symbols are in "
                           f"profiles/{PROFILE_ROUND}_*_kernel_stats.md (rocprofv3 --kernel-trace --stats of this command)",
                    by_symbol=rows[:18], class_totals=class_totals(summ, args.steps, peak))
        flops = algorithmic_flops(w["B"], w["T"], w["N"], w["S"])
        line = {
            "metric": "frames/sec (32-frame clip, 2048 pts, 512x512)", "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "Motion_Latent_Model.forward inference, B=1 x 32 frames x 2048 points x 512x512 video, "
                                   "4096 surface samples, training.frames=32, random-init weights (one clip per GPU; BASELINE configs[1]); "
                                   "the clip sits in the replayed graph's input buffers in HBM",
                       "parallelism": f"clip-parallel x{world}"},
            "launch": "eager" if fast is None else "hipGraph replay", "clips_in_flight": len(lanes),
            "comm_ranks": D.comm_ranks, "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
            "end_to_end_tflops": round(flops * world * args.steps / dt / 1e12, 1),
            "end_to_end_frac_of_bf16_peak": round(flops * args.steps / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
            "roofline": roof,
        }
        if two is not None:
            line["two_clips_in_flight"] = two
        if sus is not None:
            n, sdt = sus
            line["sustained"] = {"seconds": round(sdt, 2), "steps": n, "value": round(frames_per_step * n / sdt, 2),
                                 "unit": "frames/s", "ms_per_step": round(sdt / n * 1e3, 3)}
        blk = stages.get("stage:decoder_cross_attn_block")
        eager_blk = None
        if blk:        # the same block inside the eager, per-launch-instrumented pass (host launch gaps included)
            bms = blk["total_ms"] / args.steps
            eager_blk = {"ms_per_step": round(bms, 3), "tflops": round(blk["flops"] / args.steps / (bms * 1e-3) / 1e12, 1),
                         "frac_of_bf16_peak": round(blk["flops"] / args.steps / (bms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}
        # north_star target: >= 40 % of the dense bf16 MFMA peak on this block (reference FLOP count)
        if blk_replay is not None and "error" not in blk_replay:
            line["decoder_cross_attn_block"] = dict(blk_replay, eager_instrumented_pass=eager_blk)
        elif eager_blk is not None:
            line["decoder_cross_attn_block"] = dict(eager_blk, algorithmic_gflop=round(blk["flops"] / args.steps / 1e9, 1),
                                                    timing="eager pass, HIP events around the block", replay=blk_replay)
        import motion324_amd.Pcd_motion as pm
        import motion324_amd.transformer as tr
        from motion324_amd import switches
        line["config"]["precision_switches"] = {"M324_BF16_DECODER": int(pm.BF16_DECODER_STREAM), "M324_FUSE_HEAD": int(pm.FUSE_HEAD_N3),
                                                "M324_FOLD_LN": int(tr.FOLD_LN)}
        line["config"]["workload"] += (f"; bf16 speed mode with the decoder's residual stream in "
                                       f"{'bf16' if pm.BF16_DECODER_STREAM else 'fp32'} and the head "
                                       f"{'fused (no [rows, C] intermediate)' if pm.FUSE_HEAD_N3 else 'unfused'} "
                                       "(secondary.headline_without_precision_shortcuts has both off)")
        line["switches_non_default"] = switches.non_default()
        ref = None
        if not args.no_cpu_baseline and world == 1:
            cb, ref = cpu_baseline(sd, sample_np, w["frames"])
            line["cpu_baseline"] = cb
            err = float((out.double().cpu() - ref.double()).norm() / ref.double().norm())
            line["rel_err_vs_cpu_oracle"] = round(err, 6)
        else:
            line["cpu_baseline"] = None
        if world == 1 and fast is not None and not args.no_secondary:
            line["secondary"] = secondary_measurements(args, D, model, sd, sample, sample_np, ref, headline_ms=ms)
    return line


def run_train(args, D: Dist):
    """BASELINE configs[2] (c3, --batch 8, N = 1) and configs[3] (c4, --batch 32, N = 8): dyscene.yaml shapes."""
    import motion324_amd as m
    from motion324_amd import synth, training
    from motion324_amd.optim import FusedAdamW, backward_completion_order, cosine_with_warmup
    world, rank, dev = D.world, D.rank, D.dev
    T, N, HW = 12, 4096, 224                      # configs/dyscene.yaml: frames, num_pcd_samples = num_shape_samples, image size
    model, _ = build_model(dev, T, train=True)
    s = synth.synth_inputs(args.batch, T, N, N, HW, seed=1 + rank, with_target=True)
    sample = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
    opt = FusedAdamW(model.named_parameters(), lr=4e-4, betas=(0.9, 0.95), weight_decay=0.05, grad_clip_norm=1.0,
                     allowed_gradnorm_factor=5.0, order=backward_completion_order(model))
    m.set_precision(args.precision)
    losses, infos = [], []
    it = [0]

    def step():
        loss, _, G = training.forward_backward(model, sample, sink=opt)       # buckets leave during the backward
        opt.finish_reduce()
        info = opt.step(lr=cosine_with_warmup(it[0], 1000, 30000, 4e-4))        # 0.0 at step 0, like the reference's scheduler
        it[0] += 1
        losses.append(loss)
        infos.append(info)

    for _ in range(args.warmup):
        step()
    dt, _ = timed(D, step, args.steps)
    from motion324_amd.timing import Recorder
    rec = Recorder()
    with rec:                                     # one more step with an event pair around every GEMM / attention-forward launch
        step()
    D.fence()
    line = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        # forward FLOPs of the step (reference count); training ~ 3 x trainable forward + recompute + 1 x DINO forward
        fwd = algorithmic_flops(args.batch, T, N, N)
        line = {"metric": f"training samples/sec (dyscene.yaml shapes, batch_size_per_gpu={args.batch})",
                "value": round(world * args.batch * args.steps / dt, 3), "unit": "samples/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                "config": {"workload": f"train step: forward + backward + gradient all-reduce + fused AdamW, {args.batch} x 12 frames x "
                                       f"4096 points x 224x224 per GPU (BASELINE configs[{3 if (world > 1 or args.batch >= 32) else 2}])",
                           "parallelism": f"dp{world}", "grad_buckets": len(opt.buckets),
                           "grad_bucket_mb": round(opt.numel * 4 / len(opt.buckets) / 1e6, 1)},
                "comm_ranks": D.comm_ranks, "forward_tflop_per_step_per_gpu": round(fwd / 1e12, 2),
                "losses": [round(float(x), 6) for x in losses[-min(6, len(losses)):]],
                "grad_norm": round(infos[-1]["grad_norm"], 4), "skipped_steps": sum(1 for i in infos if i["skipped"]),
                "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                "activations": ("block internals kept in HBM while they fit half of the free memory (M324_TRAIN_STORE=1)"
                                if training.TRAIN_STORE != "0" else "checkpoint per block + recompute (M324_TRAIN_STORE=0)"),
                "roofline": roofline_from(rec, 1, args.precision, "HIP events around every GEMM / attention-forward launch of one extra "
                                          "training step (forward, dgrad, wgrad; the attention backward kernels are "
                                          f"not in the classes); symbols as in profiles/{PROFILE_ROUND}_train_p1_kernel_stats.md"),
                "cpu_baseline": None}
    return line


def run_frame_parallel(args, D: Dist):
    """BASELINE configs[4] (c5): ONE 256-frame clip (82 944 trunk tokens), frames sharded over the ranks."""
    import motion324_amd as m
    from motion324_amd import synth
    world, rank, dev = D.world, D.rank, D.dev
    T = args.frames
    model, _ = build_model(dev, T)
    from motion324_amd import parallel
    import functools
    collect = parallel.collectives_on(world)                                   # N > 1 (or M324_BENCH_COLLECT=1 on one rank: rehearsal)
    if collect:
        # every rank generates / uploads ONLY its frames of the same clip (805 MB of fp32 frames at T = 256: 7/8 of a full
        # upload would be frames the rank never reads)
        mine = parallel.partition(T, world, rank)
        s = synth.synth_inputs(1, T, 2048, args.surface, 512, seed=1, frames=mine)
        fp = functools.partial(model.forward_frame_parallel, local_frames=True, total_frames=T)
    else:
        s = synth.synth_inputs(1, T, 2048, args.surface, 512, seed=1)
        fp = None
    sample = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
    m.set_precision(args.precision)
    if args.eager:
        fast = None
    elif collect:
        # a chain of hipGraphs cut at the RCCL exchanges (graph.py _Segmenter): collectives cannot be captured on this stack
        fast = m.GraphedForward(model, forward=fp, segmented=True)
    else:
        fast = m.GraphedForward(model)
    last = [None]

    def step():
        with torch.no_grad():
            if fast is not None:
                last[0] = fast(sample).pcd_moved
            elif collect:
                last[0] = fp(sample).pcd_moved
            else:
                last[0] = model(sample).pcd_moved

    for _ in range(args.warmup):
        step()
    dt, _ = timed(D, step, args.steps)
    from motion324_amd.timing import Recorder
    rec = Recorder()
    with rec, torch.no_grad():                    # one eager forward with an event pair around every GEMM / attention launch
        if collect:
            fp(sample)
        else:
            model(sample)
    D.fence()
    line = None
    if rank == 0:
        flops = algorithmic_flops(1, T, 2048, args.surface)
        line = {"metric": f"frames/sec ({T}-frame clip, 2048 pts, 512x512, frame-parallel)", "value": round(T * args.steps / dt, 2),
                "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                "config": {"workload": f"one {T}-frame clip x 2048 points x 512x512, {args.surface} surface samples, training.frames={T} "
                                       "(BASELINE configs[4]); frames sharded over the ranks, K/V all-gather per global block",
                           "parallelism": f"frame-parallel x{world}"},
                "comm_ranks": D.comm_ranks,
                "launch": ("eager" if fast is None else "hipGraph chain cut at the K/V exchanges (collectives eager between the graphs)" if collect
                           else "hipGraph replay"),
                "video": "each rank holds only its frames" if collect else "whole clip",
                "end_to_end_tflops": round(flops * args.steps / dt / 1e12, 1),
                "end_to_end_frac_of_bf16_peak": round(flops * args.steps / dt / 1e12 / PEAK_BF16_TFLOPS / world, 4),
                "finite": bool(torch.isfinite(last[0]).all()), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                "roofline": roofline_from(rec, 1, args.precision, "HIP events around every GEMM / attention launch of one eager forward "
                                          "on the launch stream (rank 0)"),
                "cpu_baseline": None}
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "frame-parallel"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=8, help="train mode: batch_size_per_gpu (c3: 8, c4: 32)")
    ap.add_argument("--frames", type=int, default=256, help="frame-parallel mode: clip length (training.frames)")
    ap.add_argument("--surface", type=int, default=4096, help="frame-parallel mode: surface samples (the shell script uses 16384)")
    ap.add_argument("--sustain", type=float, default=3.0, help="infer mode: seconds of back-to-back replays for the `sustained` field (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-in-flight", action="store_true", help="infer mode: skip the extra two-clips-in-flight measurement")
    ap.add_argument("--no-secondary", action="store_true",
                    help="infer mode: skip the `secondary` object (c3 training step, 256-frame clip, fp32 parity mode, headline "
                         "without the precision shortcuts, H2D copy)")
    ap.add_argument("--eager", action="store_true", help="time eager per-kernel launches instead of hipGraph replay")
    ap.add_argument("--dino-streams", type=int, default=2, choices=[1, 2],
                    help="image encoder inside the hipGraph: 2 = two half-batches of frames as two branches (default), 1 = one chain (A/B)")
    ap.add_argument("--clips-in-flight", type=int, default=1, choices=[1, 2],
                    help="2: consecutive steps alternate between two HIP streams / graphs, so one clip's kernels fill the "
                         "partly filled last round of the other's (throughput mode; the default 1 is one clip at a time)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    maybe_spawn(args)                     # N > 1 without a launcher: child ranks, before any GPU call in this process
    if os.environ.get("M324_BENCH_WATCHDOG"):          # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["M324_BENCH_WATCHDOG"]), exit=True)
    D = Dist(args)
    line = {"infer": run_infer, "train": run_train, "frame-parallel": run_frame_parallel}[args.mode](args, D)
    if D.rank == 0:
        print(json.dumps(line), flush=True)
    D.close()


if __name__ == "__main__":
    main()
