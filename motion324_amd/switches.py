"""Every environment switch of the package in ONE table (name -> default, meaning), read once at import.

The product path never needs any of them: defaults are what bench.py and the tests measure.  They exist for interleaved
A/B measurements and for the parity tests that compare a fused path against its unfused form.  ``non_default()`` is what
bench.py echoes into its JSON line, so that a number measured with a switch flipped says so.

Host switches (Python; the module constants named below are initialised from here -- tests monkeypatch those constants):
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

# name: (default, module constant it initialises, meaning)
HOST: Dict[str, Tuple[str, str, str]] = {
    "M324_AUTO_GRAPH": ("1", "Pcd_motion.AUTO_GRAPH", "inference forward(): hipGraph replay from the third call with the same shapes on"),
    "M324_FUSE_HEAD": ("1", "Pcd_motion.FUSE_HEAD_N3", "bf16 inference: head Linear + GELU + Linear(C -> 3) in one GEMM epilogue (M324_AUX_N3)"),
    "M324_BF16_DECODER": ("1", "Pcd_motion.BF16_DECODER_STREAM", "bf16 inference: the decoder's two-addition residual stream in bf16 (0: fp32 like the trunk)"),
    "M324_HOIST_Q": ("1", "Pcd_motion.HOIST_DECODER_Q", "graph capture: decoder point features + q projection on the shape-encoder branch"),
    "M324_DECODE_ROWS": (str(1 << 17), "Pcd_motion.DECODE_ROWS", "max (frames x points) rows per decoder pass"),
    "M324_OVERLAP": ("1", "Pcd_motion.OVERLAP_SHAPE_ENCODER", "inference: shape encoder on a second HIP stream under the image encoder"),
    "M324_KV_OVERLAP": ("0", "Pcd_motion.KV_OVERLAP", "frame-parallel (one sample), opt-in until it has run on a multi-GPU RCCL node: 1 = every global block attends to the rank's own keys while the k|v all-gather is in flight, then to the gathered remote keys, and merges the partial softmaxes by their log-sum-exps (0: one attention after the gather)"),
    "M324_KV_REHEARSE": ("0", "Pcd_motion.KV_REHEARSE", "one rank with forced collectives (bench.py M324_BENCH_COLLECT=1): W > 1 runs every global block's overlapped form as rank 0 of W would (own keys = the first 1 / W of the frames), to price the split attention + merge on one GPU"),
    "M324_FOLD_LN": ("2", "transformer.FOLD_LN", "LayerNorm fold: 0 off, 1 bf16 streams only (the decoder), 2 every stream (trunk, DINO too)"),
    "M324_FOLD_MERGE": ("1", "transformer.FOLD_MERGE", "LayerNorm fold: the consumer GEMM merges the producer's per-block row statistics itself (0: m324_rowstats_finish launch between them)"),
    "M324_PAIR_PROJ": ("1", "transformer.PAIR_PROJ", "bf16 inference, decoder: norm_q + norm_kv in one launch and the q + k|v projections in one launch when the q projection runs inside the block (0: four launches)"),
    "M324_ATTN_BOUNDED": ("1", "transformer.ATTN_BOUNDED", "bf16 inference, long sequences: softmax without a reference maximum when the q / k RMSNorm weights bound every score (M324_ATTN_SCORES_BOUNDED)"),
    "M324_FUSE_QKV": ("1", "transformer.FUSE_QKV", "bf16 inference: q|k|v projection epilogue writes head-major Q / K / V (RMSNorm, pre-scale)"),
    "M324_FUSE_QKV_VT": ("1", "transformer.FUSE_QKV_VT", "the same for long sequences: the epilogue writes the transposed, key-permuted V"),
    "M324_TRAIN_STORE": ("1", "training.TRAIN_STORE", "training: the forward keeps block internals while they fit half of the free HBM (0: always recompute, the reference's checkpoint policy)"),
    "M324_TRAIN_DINO_FUSED": ("1", "training.DINO_FUSED", "training: the frozen DINOv2 encoder (no gradient flows through it) runs its inference form -- LayerNorm fold, fused q|k|v epilogue -- and is enqueued first in the step (0: the unfused form the trainable blocks use)"),
    "M324_DIRECT_GRADS": ("1", "backward.DIRECT_GRADS", "training: weight / bias gradients are written straight into the optimizer's flat gradient buffer (0: temporary + copy)"),
    "M324_ACC_GRADS": ("1", "backward.ACC_GRADS", "training: later gradients of a shared weight (the decoder's per-sample passes) are summed into the one it holds by the weight-gradient kernel's own reduction (0: temporary + torch add)"),
    "M324_GELU_GRAD_FWD": ("1", "backward.GELU_GRAD_FWD", "training: the fc1 GEMM of every MLP leaves gelu'(z) next to gelu(z) (M324_AUX_STORE_GELU_GRAD) and the dgrad GEMM behind fc2 multiplies by it (M324_AUX_MUL); 0: it leaves z and the dgrad epilogue evaluates erf and exp again"),
    "M324_WEIGHT_MIRROR": ("1", "optim.WEIGHT_MIRROR", "training: FusedAdamW keeps bf16 row-major and transposed copies of every Linear weight in two flat buffers, rewritten by one m324_weight_mirror launch after each update (0: Prepared casts / m324_transpose per weight and step)"),
    "M324_DEFER_COLSUM": ("1", "ops.DEFER_COLSUM", "training: the sums of the weight gradients' split-K partials and of the norm-weight partials wait in a queue and leave in one m324_colsum_multi launch per block (0: one m324_colsum launch each, at once)"),
    "M324_PRECISION": ("", "prepared.compute_dtype()", "force bf16 / fp32 (default: follow torch.autocast like the reference)"),
    "M324_LIB": ("", "lib.LIB_PATH", "path of an alternative libm324.so (lab builds)"),
    "M324_RCCL_LIB": ("", "csrc/comm.hip", "m324_comm_*: path of the RCCL library to bind (default: the copy already loaded, else librccl.so)"),
}
# Library switches (C++; read by libm324 once, when it is loaded -- csrc/runtime.hip; m324_set_tunable overrides them)
LIBRARY: Dict[str, Tuple[str, str]] = {
    "M324_GEMM": ("0", "force a GEMM schedule (v2 | v5 | v9 | v10 | v11 | v12 | v13 | v15); 0 = chooser"),
    "M324_GEMM_TN": ("0", "128: force the 128 x 128 weight-gradient kernel"),
    "M324_XCD": ("3", "tile order: bit 0 XCD-contiguous ranges (NN GEMMs; the weight-gradient GEMM's (slice, tile) items since round 6), bit 1 4 x 2 group order for wide weights, bit 2 force it; bit 3: the ring GEMMs' look-ahead pieces past the end of K fetch the last stage again (rounds 1-4) instead of nothing (A/B); bit 4: the 256 x 128 kernel (v12) without the residual prefetch (A/B)"),
    "M324_ATTN_NW": ("0", "attention forward: waves per workgroup (4 | 8); 0 = by sequence length"),
    "M324_ATTN_FLAT": ("1", "XCD-aware flat grid: 1 = global and per-frame attention, 2 = the 8-wave global attention only, 0 = 3-D grid"),
    "M324_ATTN_OCC": ("0", "attention A/B: 1 = no one-tile form, 2 = two workgroups per CU (padded LDS), 3 = the per-frame attentions (row-major V, short sequences) keep the three-stage ring, three workgroups per CU (default since round 6: two stages, four per CU)"),
    "M324_ATTN_NQ2": ("0", "attention: 64 queries per wave"),
    "M324_ATTN_BWD_NW": ("0", "attention backward: waves per workgroup -- 0 = four (the default since round 6), 8 = eight for both kernels, 84 = dQ eight + dK/dV four, 48 = the reverse, 2 = dQ with 64 queries per wave (A/B)"),
    "M324_ATTN_EXP": ("0", "attention A/B bits: 1 static priority for the younger half of an 8-wave workgroup, 2 direct stores in the one-tile form, 4 no idle-wave skip in partly filled query tiles, 8 one workgroup per frame in the shared-query one-tile form, 16 its frame-pair form with plain (not nontemporal) stores"),
    "M324_ATTN_PWG": ("1", "attention forward, long sequences: 1 = one wave per SIMD with the hand-placed stream (attention_pwg.hip), 0 = the eight-wave kernel"),
    "M324_QKV_RING": ("1", "128 x 128 chunk ring (v13) instead of the two-stage v2: bit 0 for the fused q|k|v projection (head-major epilogue), bit 1 for plain bf16 outputs (A/B)"),
    "M324_LN_ROWS": ("2", "LayerNorm: rows per wave (2 = two interleaved rows, 1 = one row: A/B)"),
    "M324_NT_MB": ("128", "GEMM: bf16 outputs (no residual) larger than this many MiB are stored nontemporal"),
    "M324_HP": ("6", "schedule v15 (one wave per SIMD, hand-placed stream, the previous tile's epilogue between the MFMAs of the current one) for K = 768 GEMMs with bf16 output and at least two 256 x 128 tiles per CU: bit 1 the bias-only / plain epilogues (the training step's projections), bit 2 the GELU epilogues from 4096 tiles on (the decoder's fc1, the 256-frame clip's), bit 0 every GELU epilogue; folded consumers it takes get the merged statistics table from the host (transformer.hp_consumer); 0: never"),
    "M324_GEMM_PERSIST": ("1", "256 x 256 chunk-ring GEMM (v10): 1 = one persistent workgroup per CU, next tile's first chunks under the epilogue; 0 = one workgroup per tile"),
}


def get(name: str) -> str:
    """value of a switch: host switches, and library switches the host mirrors (M324_HP: transformer.hp_consumer)"""
    return os.environ.get(name, (HOST.get(name) or LIBRARY[name])[0])


def flag(name: str) -> bool:
    return get(name) != "0"


def non_default() -> Dict[str, str]:
    """{switch: value} for every switch whose environment value differs from its default (host and library)."""
    out = {}
    for name, spec in list(HOST.items()) + list(LIBRARY.items()):
        v = os.environ.get(name)
        if v is not None and v.lstrip("vV") != spec[0] and v != spec[0]:
            out[name] = v
    return out


def table() -> str:
    rows = ["| switch | default | meaning |", "|---|---|---|"]
    rows += [f"| `{n}` | {d or '(unset)'} | {doc} (`{const}`) |" for n, (d, const, doc) in HOST.items()]
    rows += [f"| `{n}` | {d} | libm324: {doc} |" for n, (d, doc) in LIBRARY.items()]
    return "\n".join(rows)
