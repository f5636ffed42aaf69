"""Tower objects: checkpoint tensors -> HBM layout -> one C-ABI call per batch.

PyTorch is used for what the task allows it for — device memory, streams, H2D/D2H copies — and
nothing else; every FLOP of the forward pass is issued by libmarqo_hip.so (include/marqo_hip.h).

HBM layout (all resident for the life of the tower; a 288 GB MI355X holds every registry model at
once, so nothing is ever re-uploaded):
  * linear weights   bf16 row-major [out_features, in_features] (the MFMA "NT" operand as stored)
  * biases, LayerNorm affine, class/positional/token embeddings   fp32
  * patch-embed conv weight  bf16 [W, Kp], K = (c, ky, kx) flattened, zero-padded to a multiple of 64
  * projections      bf16 [D, W] (transposed once at load so they are plain NT GEMM operands)
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from marqo_amd import _lib as L
from marqo_amd.engine import native_queue as NQ
from marqo_amd.engine.archs import BertArch, ClipTextArch, VitArch, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD

Tensor = torch.Tensor

# upper bound of token rows pushed through one C-ABI call (bounds the scratch workspace; with
# 288 GB of HBM this is deliberately large so GEMMs see M in the 10^5 range)
MAX_ROWS_PER_CALL = 1 << 18


def _env_int(name: str, default: int, lo: int = 1) -> int:
    """integer knob from the environment; anything unparsable or below `lo` falls back to the default (a typo must not break the import)"""
    try:
        v = int(os.environ.get(name, ""))
    except ValueError:
        return default
    return v if v >= lo else default


_checked_devices: set = set()


def _require_gpu(device: str) -> torch.device:
    if not str(device).startswith("cuda"):
        raise L.MarqoHipUnavailableError(
            f"the marqo_amd engine only runs on an AMD GPU ('cuda' / 'cuda:N' device strings on ROCm); "
            f"got device={device!r}. There is no CPU fallback.")
    if not torch.cuda.is_available():
        raise L.MarqoHipUnavailableError("no GPU is visible to PyTorch-ROCm; the marqo_amd engine has no CPU fallback")
    d = torch.device(device)
    # 'cuda' -> 'cuda:<current>': tensors report an indexed device, and the towers compare devices
    d = d if d.index is not None else torch.device("cuda", torch.cuda.current_device())
    # the kernels' grids and tile orders are laid out for ONE whole MI355X (gfx950, 256 CUs in 8 XCDs): anything else is refused at load
    # (mq_check_device; the tiled GEMM launchers check again for callers of the C ABI)
    if d.index not in _checked_devices:
        lib = L.load()
        if lib.mq_check_device(int(d.index)) != L.MQ_OK:
            raise L.MarqoHipUnavailableError(lib.mq_last_error().decode())
        _checked_devices.add(d.index)
    return d


class _Holder:
    """Keeps device tensors alive and hands out raw pointers."""

    def __init__(self, device: torch.device):
        self.device = device
        self.tensors: List[Tensor] = []

    def f32(self, t: Tensor) -> int:
        d = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self.tensors.append(d)
        return d.data_ptr()

    def bf16(self, t: Tensor) -> int:
        d = t.detach().to(dtype=torch.float32).to(device=self.device).to(torch.bfloat16).contiguous()
        self.tensors.append(d)
        return d.data_ptr()

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.tensors)

    def drop(self, ptr: int) -> int:
        """forget the tensor that starts at `ptr` (its HBM goes back to the allocator once nothing else holds it) -> bytes released"""
        for i, t in enumerate(self.tensors):
            if t.data_ptr() == ptr:
                del self.tensors[i]
                return t.numel() * t.element_size()
        return 0


def _need(sd: Dict[str, Tensor], key: str, shape: Optional[Tuple[int, ...]] = None) -> Tensor:
    if key not in sd:
        raise KeyError(f"checkpoint is missing tensor '{key}'")
    t = sd[key]
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"checkpoint tensor '{key}' has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


def _head_dim(width: int, heads: int) -> int:
    """model head dim; the attention kernel runs 64- / 96- / 112- / 128-wide heads, anything else is zero-padded at load (_pad_heads)"""
    if heads < 1 or width % heads:
        raise ValueError(f"width {width} is not divisible by heads {heads}")
    d = width // heads
    if d > 128:
        raise ValueError(f"attention head dim must be <= 128 for the gfx950 attention kernel (width={width}, heads={heads}: {d})")
    return d


KERNEL_HEAD_DIMS = (64, 96, 112, 128)  # head strides csrc/attention.hip is instantiated for


def _kernel_head_dim(d: int, heads: int = 2) -> int:
    """head width the kernel runs for a model head dim d: the smallest instantiated stride >= d whose attention width heads * hp
    keeps the GEMM's K a multiple of 64: 32 / 16 -> 64 (e5-small, MiniLM), 80 / 88 -> 96 (ViT-H / g), 104 -> 112 (ViT-bigG)"""
    for hp in KERNEL_HEAD_DIMS:
        if hp >= d and (heads * hp) % 64 == 0:
            return hp
    return 128


def _pad_heads(qkv_w: Tensor, qkv_b: Tensor, out_w: Tensor, heads: int, d: int) -> Tuple[Tensor, Tensor, Tensor]:
    """[3W, W] / [3W] / [W, W] with d-wide heads -> [3*heads*hp, W] / [3*heads*hp] / [W, heads*hp] (hp = _kernel_head_dim): each
    head's Q / K / V rows and out-projection columns are zero-padded to hp (zero key / query dims add nothing to q.k, zero value
    dims meet zero out-proj columns), and Q is scaled by sqrt(hp / d) so that the kernel's 1/sqrt(hp) softmax scale equals the
    model's 1/sqrt(d)."""
    W = out_w.shape[0]
    hp = _kernel_head_dim(d, heads)
    q, k, v = qkv_w.float().view(3, heads, d, W).unbind(0)
    qb, kb, vb = qkv_b.float().view(3, heads, d).unbind(0)
    sc = (float(hp) / d) ** 0.5
    pad_w = lambda t: torch.nn.functional.pad(t, (0, 0, 0, hp - d)).reshape(heads * hp, W)
    pad_b = lambda t: torch.nn.functional.pad(t, (0, hp - d)).reshape(heads * hp)
    qkv_w2 = torch.cat([pad_w(q * sc), pad_w(k), pad_w(v)], 0)
    qkv_b2 = torch.cat([pad_b(qb * sc), pad_b(kb), pad_b(vb)], 0)
    out_w2 = torch.nn.functional.pad(out_w.float().view(W, heads, d), (0, hp - d)).reshape(W, heads * hp)
    return qkv_w2, qkv_b2, out_w2


def _ceil64(v: int) -> int:
    return (v + 63) // 64 * 64


def _pad_mlp(fc1_w: Tensor, fc1_b: Tensor, fc2_w: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """MLP hidden sizes that are not a multiple of 64 (ViT-SO400M: 4304) are zero-padded: the extra hidden units are act(0 + 0) = 0
    for GELU / QuickGELU and meet zero fc2 columns — exact."""
    F = fc1_w.shape[0]
    Fp = _ceil64(F)
    if Fp == F:
        return fc1_w, fc1_b, fc2_w
    pad = torch.nn.functional.pad
    return pad(fc1_w.detach().float(), (0, 0, 0, Fp - F)), pad(fc1_b.detach().float(), (0, Fp - F)), pad(fc2_w.detach().float(), (0, Fp - F))


def _encoder_cfg(width, layers, heads, mlp_dim, quick_gelu, post_ln, mask, eps) -> L.EncoderCfg:
    mlp_dim = _ceil64(mlp_dim)
    d = _head_dim(width, heads)
    hp = _kernel_head_dim(d, heads)
    return L.EncoderCfg(width=width, layers=layers, heads=heads, mlp_dim=mlp_dim,
                        act=L.MQ_ACT_QUICKGELU if quick_gelu else L.MQ_ACT_GELU,
                        post_ln=1 if post_ln else 0, mask=mask, ln_eps=eps, precision=L.MQ_PREC_BF16,
                        attn_width=0 if d == hp else heads * hp, fp8_first_layer=0, mlp_glu=0, d_fp8_act_scale=None, d_fp8_act_amax=None,
                        d_rope_inv_freq=None)


class _Fp8State:
    """fp8 (e4m3) side of an encoder: per-channel-quantised copies of the four block GEMM weights and the static
    per-tensor activation scales [layers, 2] = (attention output, MLP hidden) with their calibration accumulator."""

    def __init__(self, lib, holder: "_Holder", blocks, layers: int, W: int, F: int, device: torch.device, Wa: Optional[int] = None):
        Wa = Wa or W  # attention width (heads * kernel head dim)
        self.scale = torch.full((layers, 2), 16.0 / 448.0, dtype=torch.float32, device=device)  # pre-calibration guess
        self.amax = torch.zeros(layers, 2, dtype=torch.float32, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        for i in range(layers):
            b = blocks[i]
            for name, (n, k) in (("qkv", (3 * Wa, W)), ("out", (W, Wa)), ("fc1", (F, W)), ("fc2", (W, F))):
                w8 = torch.empty(n, k, dtype=torch.uint8, device=device)
                ws = torch.empty(n, dtype=torch.float32, device=device)
                L.check(lib.mq_quantize_weights_fp8(getattr(b, name + "_w"), k, w8.data_ptr(), k, ws.data_ptr(), n, k, stream),
                        "mq_quantize_weights_fp8")
                holder.tensors += [w8, ws]
                setattr(b, name + "_w8", w8.data_ptr())
                setattr(b, name + "_ws", ws.data_ptr())
        self.calibrated = False

    def attach(self, enc: L.EncoderCfg, calibrating: bool) -> None:
        enc.precision = L.MQ_PREC_FP8
        enc.d_fp8_act_scale = self.scale.data_ptr()
        enc.d_fp8_act_amax = self.amax.data_ptr() if calibrating else None

    def fold(self, margin: float = 1.0) -> None:
        """scale <- observed amax * margin / 448 (entries never observed keep their value); reset the accumulator"""
        seen = self.amax > 0
        self.scale[seen] = (self.amax[seen] * (margin / 448.0))
        self.amax.zero_()


# LayerNorm folding of the pre-LN CLIP blocks (csrc/gemm_epilogue.h, MQ_EPI_LN_APPLY): the loaders prepare gamma-folded copies of the QKV / fc1
# weights (+ folded bias and column sums); on the bf16 residual stream the tiled GEMMs then read the stream itself and no LayerNorm kernel runs in
# front of them.  The un-folded weights stay for the small-call kernels (fused-LayerNorm skinny GEMMs) and the fp32-stream towers.
# MARQO_AMD_LN_FOLD=0: do not build the folded tensors (the towers then always launch their LayerNorms).
LN_FOLD = os.environ.get("MARQO_AMD_LN_FOLD", "1") != "0"


# per-block tensor names: open_clip ResidualAttentionBlock / timm Block (the SigLIP trunks)
_OPEN_CLIP_KEYS = dict(block="resblocks.{}.", ln1="ln_1", qkv_w="attn.in_proj_weight", qkv_b="attn.in_proj_bias", out="attn.out_proj",
                       ln2="ln_2", fc1="mlp.c_fc", fc2="mlp.c_proj")
_TIMM_KEYS = dict(block="blocks.{}.", ln1="norm1", qkv_w="attn.qkv.weight", qkv_b="attn.qkv.bias", out="attn.proj",
                  ln2="norm2", fc1="mlp.fc1", fc2="mlp.fc2")


def _clip_blocks(h: _Holder, sd, prefix: str, layers: int, W: int, F: int, heads: int, keys=_OPEN_CLIP_KEYS):
    d = _head_dim(W, heads)
    padded = d != _kernel_head_dim(d, heads)  # ViT-H / g / bigG: 80 / 88 / 104-wide heads run as 96 / 96 / 112
    arr = (L.BlockWeights * layers)()
    k = keys
    for i in range(layers):
        p = prefix + k["block"].format(i)
        b = arr[i]
        b.ln1_g = h.f32(_need(sd, p + k["ln1"] + ".weight", (W,)))
        b.ln1_b = h.f32(_need(sd, p + k["ln1"] + ".bias", (W,)))
        qkv_w, qkv_b = _need(sd, p + k["qkv_w"], (3 * W, W)), _need(sd, p + k["qkv_b"], (3 * W,))
        out_w = _need(sd, p + k["out"] + ".weight", (W, W))
        if padded:
            qkv_w, qkv_b, out_w = _pad_heads(qkv_w.detach(), qkv_b.detach(), out_w.detach(), heads, d)
        b.qkv_w, b.qkv_b, b.out_w = h.bf16(qkv_w), h.f32(qkv_b), h.bf16(out_w)
        b.out_b = h.f32(_need(sd, p + k["out"] + ".bias", (W,)))
        b.ln2_g = h.f32(_need(sd, p + k["ln2"] + ".weight", (W,)))
        b.ln2_b = h.f32(_need(sd, p + k["ln2"] + ".bias", (W,)))
        fc1_w, fc1_b, fc2_w = _pad_mlp(_need(sd, p + k["fc1"] + ".weight", (F, W)), _need(sd, p + k["fc1"] + ".bias", (F,)),
                                       _need(sd, p + k["fc2"] + ".weight", (W, F)))
        b.fc1_w, b.fc1_b, b.fc2_w = h.bf16(fc1_w), h.f32(fc1_b), h.bf16(fc2_w)
        b.fc2_b = h.f32(_need(sd, p + k["fc2"] + ".bias", (W,)))
        if LN_FOLD:
            # LayerNorm folding (csrc/gemm_epilogue.h): LN(x) @ W^T = rstd * (x @ (g*W)^T - mean * colsum(g*W)) + (b + W @ beta).
            # colsum is taken over the bf16-ROUNDED folded weight (what the MFMA multiplies), the bias in fp32 from the fp32 W; the (possibly
            # head- / MLP-padded) tensors the block really runs are the ones folded.
            for name, w_t, b_t, lg in (("qkv", qkv_w, qkv_b, k["ln1"]), ("fc1", fc1_w, fc1_b, k["ln2"])):
                w32 = w_t.detach().to(torch.float32)
                gam, bet = sd[p + lg + ".weight"].detach().to(torch.float32), sd[p + lg + ".bias"].detach().to(torch.float32)
                wf = (w32 * gam.unsqueeze(0)).to(torch.bfloat16)
                setattr(b, name + "_wf", h.bf16(wf))
                setattr(b, name + "_sf", h.f32(wf.to(torch.float32).sum(dim=1)))
                setattr(b, name + "_bf", h.f32(b_t.detach().to(torch.float32) + w32 @ bet))
    return arr


# EVA02 blocks: up * silu(gate) in the (up | gate) GEMM's epilogue (MQ_EPI_GLU; 0 = the round-5 form: the GEMM writes (up | gate), glu_ln_kernel multiplies)
EVA_GLU_EPILOGUE = os.environ.get("MARQO_AMD_EVA_GLU_EPILOGUE", "1") != "0"


def _eva_blocks(h: _Holder, sd, prefix: str, layers: int, W: int, F: int, heads: int):
    """timm EvaBlock tensors (eva.py; `blocks.{i}.`): norm1, attn.{q_proj, k_proj (no bias), v_proj} or the fused attn.qkv + q_bias / v_bias,
    attn.norm (the LayerNorm in front of attn.proj; absent without `scale_attn_inner`), attn.proj, norm2, mlp.{fc1_g, fc1_x, norm, fc2}
    (timm SwiGLU: fc2(norm(silu(fc1_g(x)) * fc1_x(x)))).  fc1 is stored as (up, gate) = (fc1_x, fc1_g) rows interleaved 16 by 16, the hidden width F zero-padded
    to a multiple of 64: silu(0) * 0 = 0 meets zero LayerNorm weights and zero fc2 columns — exact; the statistics run over F (mlp_ln_dim)."""
    if _head_dim(W, heads) != _kernel_head_dim(_head_dim(W, heads), heads):
        raise ValueError("EVA02 towers with heads that are not 64 / 96 / 112 / 128 wide are not runnable (rotary positions on padded heads)")
    Fp = _ceil64(F)
    pad = torch.nn.functional.pad
    arr = (L.BlockWeights * layers)()
    for i in range(layers):
        p = prefix + f"blocks.{i}."
        f32 = lambda k, shape: _need(sd, p + k, shape).detach().to(torch.float32)
        b = arr[i]
        b.ln1_g, b.ln1_b = h.f32(f32("norm1.weight", (W,))), h.f32(f32("norm1.bias", (W,)))
        if p + "attn.qkv.weight" in sd:
            qkv_w = f32("attn.qkv.weight", (3 * W, W))
            qb = f32("attn.q_bias", (W,)) if p + "attn.q_bias" in sd else torch.zeros(W)
            vb = f32("attn.v_bias", (W,)) if p + "attn.v_bias" in sd else torch.zeros(W)
        else:
            qkv_w = torch.cat([f32("attn.q_proj.weight", (W, W)), f32("attn.k_proj.weight", (W, W)), f32("attn.v_proj.weight", (W, W))], dim=0)
            qb = f32("attn.q_proj.bias", (W,)) if p + "attn.q_proj.bias" in sd else torch.zeros(W)
            vb = f32("attn.v_proj.bias", (W,)) if p + "attn.v_proj.bias" in sd else torch.zeros(W)
        qkv_b = torch.cat([qb, torch.zeros(W), vb])                                        # (keys carry no bias)
        b.qkv_w, b.qkv_b = h.bf16(qkv_w), h.f32(qkv_b)
        if p + "attn.norm.weight" in sd:
            b.attn_ln_g, b.attn_ln_b = h.f32(f32("attn.norm.weight", (W,))), h.f32(f32("attn.norm.bias", (W,)))
        b.out_w, b.out_b = h.bf16(f32("attn.proj.weight", (W, W))), h.f32(f32("attn.proj.bias", (W,)))
        b.ln2_g, b.ln2_b = h.f32(f32("norm2.weight", (W,))), h.f32(f32("norm2.bias", (W,)))
        up_w, up_b = f32("mlp.fc1_x.weight", (F, W)), f32("mlp.fc1_x.bias", (F,))
        gate_w, gate_b = f32("mlp.fc1_g.weight", (F, W)), f32("mlp.fc1_g.bias", (F,))
        # (up, gate) rows interleaved 16 by 16 (mq_encoder_cfg.mlp_glu = 2): a lane of the GEMM's epilogue then holds up AND gate of the same hidden
        # units and forms up * silu(gate) itself (MQ_EPI_GLU) — the (up | gate) tensor is never written
        il = (lambda u, g_: torch.stack([u.reshape(Fp // 16, 16, *u.shape[1:]), g_.reshape(Fp // 16, 16, *g_.shape[1:])], dim=1).reshape(2 * Fp, *u.shape[1:])) \
            if EVA_GLU_EPILOGUE else (lambda u, g_: torch.cat([u, g_], dim=0))
        fc1_w = il(pad(up_w, (0, 0, 0, Fp - F)), pad(gate_w, (0, 0, 0, Fp - F)))
        fc1_b = il(pad(up_b, (0, Fp - F)), pad(gate_b, (0, Fp - F)))
        b.fc1_w, b.fc1_b = h.bf16(fc1_w), h.f32(fc1_b)
        if p + "mlp.norm.weight" in sd:
            b.mlp_ln_g, b.mlp_ln_b = h.f32(pad(f32("mlp.norm.weight", (F,)), (0, Fp - F))), h.f32(pad(f32("mlp.norm.bias", (F,)), (0, Fp - F)))
        b.fc2_w, b.fc2_b = h.bf16(pad(f32("mlp.fc2.weight", (W, F)), (0, Fp - F))), h.f32(f32("mlp.fc2.bias", (W,)))
        if LN_FOLD:   # norm1 into the QKV GEMM, norm2 into the (up | gate) GEMM (as _clip_blocks)
            for name, w32, b32, lg in (("qkv", qkv_w, qkv_b, "norm1"), ("fc1", fc1_w, fc1_b, "norm2")):
                wf = (w32 * f32(lg + ".weight", (W,)).unsqueeze(0)).to(torch.bfloat16)
                setattr(b, name + "_wf", h.bf16(wf))
                setattr(b, name + "_sf", h.f32(wf.to(torch.float32).sum(dim=1)))
                setattr(b, name + "_bf", h.f32(b32 + w32 @ f32(lg + ".bias", (W,))))
            # ... and the sub-LayerNorms into the GEMMs behind them (ABI 12, csrc/towers.hip block_eva): attn.norm into attn.proj, mlp.norm into mlp.fc2 (over
            # the padded hidden width: zero LayerNorm weights meet zero fc2 columns).  Their rows' statistics come from the attention kernel / the gated epilogue.
            subs = []
            if p + "attn.norm.weight" in sd:
                subs.append(("out", f32("attn.proj.weight", (W, W)), f32("attn.proj.bias", (W,)), f32("attn.norm.weight", (W,)), f32("attn.norm.bias", (W,))))
            if p + "mlp.norm.weight" in sd and EVA_GLU_EPILOGUE:
                subs.append(("fc2", pad(f32("mlp.fc2.weight", (W, F)), (0, Fp - F)), f32("mlp.fc2.bias", (W,)), pad(f32("mlp.norm.weight", (F,)), (0, Fp - F)),
                             pad(f32("mlp.norm.bias", (F,)), (0, Fp - F))))
            for name, w32, b32, gam, bet in subs:
                wf = (w32 * gam.unsqueeze(0)).to(torch.bfloat16)
                setattr(b, name + "_wf", h.bf16(wf))
                setattr(b, name + "_sf", h.f32(wf.to(torch.float32).sum(dim=1)))
                setattr(b, name + "_bf", h.f32(b32 + w32 @ bet))
    return arr


# Single-request calls (the search path: one query text / one image per vectorise()) are ~75-150 dependent launches of 5-10 us.
# The launch sequence of such a call depends only on (tower, token count), so it is captured once per shape in a hipGraph
# (torch.cuda.CUDAGraph over the stream the C ABI enqueues on) with static input / output / workspace buffers and replayed.
# Measured (profiles/r01f_latency.txt): p50 -3..-4 %, p95 -15 % (0.81 -> 0.68 ms, CLIP text B/32): the host enqueue was already
# hidden behind the GPU; what remains is the GPU-side latency of ~75 dependent small kernels (~9 us each).
GRAPHS = os.environ.get("MARQO_AMD_GRAPHS", "1") != "0"
MAX_GRAPHS_PER_TOWER = 192   # each entry owns a small workspace; calls of rarer shapes launch eagerly


class _GraphedCall:
    """One captured launch sequence: `launch()` must enqueue on the current stream and touch only `inp`, `out` and buffers it owns."""

    def __init__(self, device: torch.device, inp: Tensor, out: Tensor, keep: tuple, launch) -> None:
        self.inp, self.out, self._keep = inp, out, keep
        launch()                                   # eager once: lazy one-time host work (kernel attributes) happens outside the capture
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other request threads keep allocating / launching on their own streams while this one captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            launch()

        self._last: Optional[torch.cuda.Event] = None

    def __call__(self, src: Tensor) -> Tensor:
        """(called under the tower's lock) callers may sit on different streams: order this use of the static buffers behind the
        previous one on the GPU, not just on the host"""
        cur = torch.cuda.current_stream(self.inp.device)
        if self._last is not None:
            cur.wait_event(self._last)
        self.inp.copy_(src, non_blocking=True)
        self.graph.replay()
        if getattr(_request_tls, "host_output", False):
            # the loaders hand the embedding to the host anyway (vectorise() returns lists / ndarrays): one D2H copy straight out of the
            # static buffer instead of a device clone + the caller's .cpu()
            out = torch.empty(self.out.shape, dtype=self.out.dtype, pin_memory=True)
            out.copy_(self.out, non_blocking=True)
            self._last = torch.cuda.Event()
            self._last.record(cur)
            self._last.synchronize()
            return out
        out = self.out.clone()                     # the static output is overwritten by the next replay
        self._last = torch.cuda.Event()
        self._last.record(cur)
        return out


# Every request thread enqueues on its own HIP stream (MARQO_AMD_THREAD_STREAMS=0: the caller's current stream), so that one
# request's pinned H2D copy / host packing overlaps another request's kernels instead of queueing behind them on the shared
# default stream.
THREAD_STREAMS = os.environ.get("MARQO_AMD_THREAD_STREAMS", "1") != "0"
_request_tls = threading.local()


class request_stream:
    """`with request_stream(device, device_output):` — run the body on the calling thread's private stream.  Inputs produced on the
    caller's current stream are ordered before it; outputs are handed back either synchronised (the loaders' D2H copy) or, for
    device-resident results (`device_output=True`), ordered before the caller's stream continues."""

    def __init__(self, device, device_output: bool = False):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.device_output = device_output
        self._ctx = None

    def __enter__(self):
        if not getattr(_request_tls, "depth", 0):
            _request_tls.host_output = not self.device_output
        if not THREAD_STREAMS or not torch.cuda.is_available() or self.device.type != "cuda":
            return self
        streams = getattr(_request_tls, "streams", None)
        if streams is None:
            streams = _request_tls.streams = {}
        if getattr(_request_tls, "depth", 0):      # nested (encode_image inside encode): already on the private stream
            _request_tls.depth += 1
            self._nested = True
            return self
        self._nested = False
        key = (self.device.type, self.device.index if self.device.index is not None else torch.cuda.current_device())
        s = streams.get(key)
        if s is None:
            s = streams[key] = torch.cuda.Stream(torch.device(*key))
        self._outer = torch.cuda.current_stream(torch.device(*key))
        s.wait_stream(self._outer)
        self._s = s
        self._ctx = torch.cuda.stream(s)
        self._ctx.__enter__()
        _request_tls.depth = 1
        return self

    def __exit__(self, *exc):
        if getattr(self, "_nested", False):
            _request_tls.depth -= 1
            return False
        _request_tls.host_output = False
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
            _request_tls.depth = 0
            if self.device_output:
                self._outer.wait_stream(self._s)
        return False


# Experiment knob (MARQO_AMD_CHAIN_LARGE_CALLS=1, off by default): chain chip-filling tower calls of concurrent request threads on the GPU
# instead of letting their kernels interleave.  A call of >= LARGE_CALL_ROWS token rows launches kernels that each fill the 256 CUs, so the
# idea was that FIFO execution (each such call waits, on the GPU, for the completion event of the previous one on that device, while its own
# pack / H2D / resize still overlap) keeps cache residency.  Measured with 256-image callers (profiles/r02z_chain_large_calls_ab.txt):
# 2 callers 70.0 k vs 67.6 k embeddings/s unchained (+3 %), 4 callers 61 k vs 68-73 k (-10 %): the hardware's own interleaving of four
# streams hides the tails of one call behind another's kernels better than strict FIFO does.
CHAIN_LARGE_CALLS = os.environ.get("MARQO_AMD_CHAIN_LARGE_CALLS", "0") == "1"
LARGE_CALL_ROWS = int(os.environ.get("MARQO_AMD_LARGE_CALL_ROWS", "4096"))
_chain_lock = threading.Lock()
_chain_last: Dict[int, "torch.cuda.Event"] = {}


class _large_call:
    """`with _large_call(device, rows):` around the enqueue of one tower call"""

    def __init__(self, device: torch.device, rows: int):
        self.on = CHAIN_LARGE_CALLS and rows >= LARGE_CALL_ROWS and not torch.cuda.is_current_stream_capturing()
        self.device = device

    def __enter__(self):
        if self.on:
            idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
            self._idx, self._mine = idx, torch.cuda.Event()
            with _chain_lock:
                prev, _chain_last[idx] = _chain_last.get(idx), self._mine
            self._stream = torch.cuda.current_stream(self.device)
            if prev is not None:
                self._stream.wait_event(prev)
        return self

    def __exit__(self, *exc):
        if self.on:
            self._mine.record(self._stream)   # (recorded even when the body raised: later callers must never wait on an unrecorded event)
        return False


class _TowerBase:
    _fp8: Optional[_Fp8State] = None

    def _enable_fp8(self, blocks, layers: int, W: int, F: int) -> None:
        if W % 128 or F % 128:
            raise ValueError(f"the fp8 path needs width / mlp_dim multiples of 128 (got {W}, {F})")
        with torch.cuda.device(self.device):
            self._fp8 = _Fp8State(self.lib, self._h, blocks, layers, W, F, self.device, Wa=self.cfg.enc.attn_width or W)
            self._fp8.attach(self.cfg.enc, calibrating=False)

    def calibrate_fp8(self, run, passes: int = 2, margin: float = 1.0) -> None:
        """Static activation-scale calibration: `run()` must push a representative batch through this tower.  Each pass
        records max|activation| of the two fp8 activation tensors per layer on the device and folds it into the scales
        (2 passes: the first one runs on the pre-calibration guess).  Afterwards the scales are frozen (deterministic)."""
        if self._fp8 is None:
            raise RuntimeError("tower was not built with precision='fp8'")
        self._fp8.calibrated = False  # (single-request graphs are not replayed while the recording passes run)
        for _ in range(passes):
            self._fp8.attach(self.cfg.enc, calibrating=True)
            run()
            torch.cuda.synchronize(self.device)
            self._fp8.fold(margin)
        self._fp8.attach(self.cfg.enc, calibrating=False)
        self._fp8.calibrated = True

    # ---- residual-stream policy -----------------------------------------------------------------------------------------------
    # Keeping x in bf16 between the blocks of a pre-LN tower halves the bytes of every residual epilogue and LayerNorm: +3.6 % (ViT-B/32),
    # +7 % (ViT-L/14), +13 % (CLIP text B/32) embeddings/s (profiles/r03f_residual_ln_ab.txt).  What it costs is MODEL dependent: 4.7e-5 ..
    # 1.2e-4 of cosine error on the registry-shaped fixtures, but 7.8e-3 on the SigLIP-small golden text tower, whose residual stream carries
    # values two orders of magnitude above its per-block updates (8 mantissa bits then drop most of an update).  So it is decided per model,
    # deterministically at load, like the fp8 split: the fixed seeded calibration batch runs through both forms and bf16 is kept only when
    # max (1 - cos) against the fp32 stream is within the budget.  MARQO_AMD_RESIDUAL_STREAM = auto (default) | fp32 | bf16.
    # Budget 5e-4 (MARQO_AMD_RESIDUAL_STREAM_BUDGET): trained-like fixtures measure 1.5e-4 .. 4.2e-4 on the calibration batch and 7e-5 ..
    # 2.8e-4 against the fp32 CPU oracle on held-out inputs (profiles/r03h_residual_policy_report.txt) — inside the 1e-3 north-star
    # tolerance with a factor of three to spare; the SigLIP-small golden text tower (7.8e-3) is refused.
    RESIDUAL_STREAM_BUDGET = float(os.environ.get("MARQO_AMD_RESIDUAL_STREAM_BUDGET", "5e-4"))
    residual_stream: str = "fp32"
    residual_stream_error: Optional[float] = None

    def tune_residual_stream(self, run, budget: Optional[float] = None) -> str:
        """`run()` pushes the tower's fixed calibration batch through it and returns the [n, D] embeddings -> 'bf16' | 'fp32'"""
        enc = self.cfg.enc
        getattr(self, "_graphs", {}).clear()   # captured single-item graphs bake in the stream's layout and launch sequence: a re-tune must not replay stale ones
        mode = os.environ.get("MARQO_AMD_RESIDUAL_STREAM", "auto").lower()
        if self.precision != "bf16" or mode == "fp32":
            enc.residual_stream, self.residual_stream = 2, "fp32"
            return self.residual_stream
        if mode == "bf16":
            enc.residual_stream, self.residual_stream = 1, "bf16"
            return self.residual_stream
        budget = self.RESIDUAL_STREAM_BUDGET if budget is None else float(budget)
        enc.residual_stream = 2
        ref = run().double()
        enc.residual_stream = 1
        out = run().double()
        cos = (out * ref).sum(-1) / (out.norm(dim=-1) * ref.norm(dim=-1))
        e = float((1 - cos).max())
        self.residual_stream_error = e
        ok = e <= budget and bool(torch.isfinite(out).all())
        enc.residual_stream, self.residual_stream = (1, "bf16") if ok else (2, "fp32")
        import logging
        logging.getLogger(__name__).info("residual stream: %s (bf16 vs fp32 stream on the calibration batch: 1 - cos %.2e, budget %.1e)",
                                         self.residual_stream, e, budget)
        return self.residual_stream

    # ---- fp8 policy --------------------------------------------------------------------------------------------------------
    # e4m3 operands carry ~2.6 % rms relative rounding noise each, whatever the scaling granularity (per tensor, per row or MX
    # blocks: oracle/fp8_sim.py, tests/studies/fp8_numerics_study.py -> profiles/r02_fp8_numerics_sim.txt): a GEMM output is ~3.7 % noise, a 24-block
    # ViT-L/14 with every block on fp8 ends at 1 - cos = 2e-3 on benign weights and 5e-3 on trained-like ones — outside the 1e-3
    # north-star tolerance.  Noise injected in early blocks is amplified by all later ones, so the policy keeps the FIRST blocks on
    # bf16 operands and runs the LAST ones on fp8: `tune_fp8` measures, on a fixed seeded calibration batch at load, the error of
    # each split against the tower's own bf16 output and keeps the most fp8 blocks that stay inside the budget.
    # max (1 - cos) on the calibration batch.  5e-4 since round 4 (7e-4 before): with every default policy stacked, the trained-like 24-block
    # ViT-L/14 at the 7e-4 budget (6.8e-4 at load) measured 1.03e-3 against the fp32 oracle on 260 HELD-OUT crops of natural-image statistics —
    # the calibration batch under-states held-out error by ~1.5x (tests/test_fp8_gpu.py::test_default_policies_together_...); 5e-4 keeps the
    # held-out figure inside the 1e-3 north-star tolerance with that factor applied
    FP8_BUDGET = float(os.environ.get("MARQO_AMD_FP8_BUDGET", "5e-4"))
    FP8_STREAM_SHARE = 0.25  # an fp8 tower takes the bf16 residual stream when that alone costs <= this share of the budget
    FP8_SCALE_MARGIN = 2.0   # static activation scales = calibration amax x margin / 448: one binade of head-room for unseen inputs
    fp8_first_layer: int = 0
    fp8_mlp_extra: int = 0
    fp8_policy_trace: Optional[list] = None
    fp8_calibration_error: Optional[float] = None
    fp8_all_blocks_error: Optional[float] = None

    def tune_fp8(self, run, budget: Optional[float] = None, margin: Optional[float] = None) -> int:
        """Deterministic load-time calibration of an fp8 tower on the caller's fixed calibration batch (`run()` pushes it through this
        tower and returns the [n, D] embeddings): (1) static activation scales from two recording passes with every block on fp8,
        x `margin`; (2) the split `fp8_first_layer` = the smallest first fp8 block whose max (1 - cos) against the bf16 run of the same
        batch is <= budget (binary search: the error grows as the split moves towards block 0).  Returns the split."""
        if self._fp8 is None:
            raise RuntimeError("tower was not built with precision='fp8'")
        budget = self.FP8_BUDGET if budget is None else float(budget)
        enc, layers = self.cfg.enc, self.cfg.enc.layers
        getattr(self, "_graphs", {}).clear()   # (as in tune_residual_stream: graphs captured under the previous policy would disagree with the eager path)
        enc.fp8_first_layer, enc.fp8_mlp_extra = 0, 0
        self.calibrate_fp8(run, passes=2, margin=self.FP8_SCALE_MARGIN if margin is None else margin)
        self._fp8.calibrated = False           # no graph capture while the split is being searched
        enc.precision = L.MQ_PREC_BF16
        ref = run().double()
        enc.precision = L.MQ_PREC_FP8

        def err_split(first: int, extra: int) -> float:
            enc.fp8_first_layer, enc.fp8_mlp_extra = first, extra
            out = run().double()
            cos = (out * ref).sum(-1) / (out.norm(dim=-1) * ref.norm(dim=-1))
            return float((1 - cos).max())

        def err(first: int) -> float:
            enc.fp8_first_layer, enc.fp8_mlp_extra = first, 0
            out = run().double()
            cos = (out * ref).sum(-1) / (out.norm(dim=-1) * ref.norm(dim=-1))
            return float((1 - cos).max())
        # The stream itself (pre-LN towers): bf16 rows halve the residual traffic of EVERY block, e4m3 or not.  Its rounding spends part of the
        # same budget (all errors below are measured against the fp32-stream bf16 run), so it is taken only when it costs at most
        # FP8_STREAM_SHARE of the budget with every block still on bf16 operands.
        enc.residual_stream, self.residual_stream, self.residual_stream_error = 2, "fp32", None
        mode = os.environ.get("MARQO_AMD_RESIDUAL_STREAM", "auto").lower()
        if not enc.post_ln and mode != "fp32":
            enc.residual_stream = 1
            e_stream = err_split(layers, 0)
            if mode == "bf16" or e_stream <= self.FP8_STREAM_SHARE * budget:       # (NaN compares false)
                self.residual_stream, self.residual_stream_error = "bf16", e_stream
            else:
                enc.residual_stream = 2
        e0 = err(0)
        self.fp8_all_blocks_error = e0
        if e0 <= budget:
            first, e = 0, e0
        else:
            lo, hi, e = 0, layers, 0.0          # err(lo) > budget, err(hi) <= budget (hi == layers: every block bf16)
            while hi - lo > 1:
                mid = (lo + hi) // 2
                em = err(mid)
                if em <= budget:
                    hi, e = mid, em
                else:
                    lo = mid
            first = hi
        # Per-GEMM-type refinement (pre-LN towers): a block in front of the split may run only its MLP half on e4m3 (fp8_mlp_extra: two
        # thirds of the block's GEMM FLOPs for the noise of two of its four GEMMs).  Two-dimensional search at load: for a few candidate
        # splits s >= `first`, the largest `extra` <= s whose error stays inside the budget (binary search: the error grows with extra);
        # keep the (split, extra) with the largest e4m3 share of the GEMM FLOPs = (layers - s) + 2/3 extra.
        extra = 0
        self.fp8_policy_trace = [(first, 0, e)]
        if not enc.post_ln and os.environ.get("MARQO_AMD_FP8_MLP_ONLY", "1") != "0":
            best = (self._fp8_share(layers, first, 0), first, 0, e)
            step = max(1, layers // 8)
            for s_ in sorted({min(layers, first + k * step) for k in range(0, 9)} | {layers}):
                if s_ == 0:
                    continue
                lo_x, hi_x = 0, s_                       # err(s_, lo_x) <= budget is known for lo_x = 0 (s_ >= first)
                e_s = err_split(s_, 0) if s_ != first else e
                if e_s > budget:
                    continue
                e_at = {0: e_s}
                while hi_x - lo_x > 0:
                    mid = (lo_x + hi_x + 1) // 2
                    em = err_split(s_, mid)
                    e_at[mid] = em
                    if em <= budget:
                        lo_x = mid
                    else:
                        hi_x = mid - 1
                share = self._fp8_share(layers, s_, lo_x)
                self.fp8_policy_trace.append((s_, lo_x, e_at[lo_x]))
                if share > best[0] + 1e-9:
                    best = (share, s_, lo_x, e_at[lo_x])
            _, first, extra, e = best
        enc.fp8_first_layer, enc.fp8_mlp_extra = first, extra
        self.fp8_first_layer, self.fp8_mlp_extra, self.fp8_calibration_error = first, extra, e
        self._fp8.calibrated = True
        import logging
        logging.getLogger(__name__).info("fp8 policy: blocks [%d, %d) on e4m3 + the MLP halves of blocks [%d, %d), %s residual stream, 1 - cos vs bf16 on "
                                         "the calibration batch %.2e (all blocks: %.2e, budget %.1e)", first, layers, first - extra, first,
                                         self.residual_stream, e, e0, budget)
        return first

    def release_unused_folded(self) -> int:
        """the gamma-folded QKV / fc1 copies (+ bias, column sums: 7/12 of a block's weight bytes again, ~2 GB on ViT-bigG) are read only by
        blocks that can take the folded path: bf16 residual stream, block in front of the fp8 split (fc1: also in front of the MLP-only fp8 blocks).
        Called by the loaders once the load-time policies are fixed: every other block's folded tensors are freed and its pointers nulled (towers.hip's
        fold_ok then takes the LayerNorm path, as with MARQO_AMD_LN_FOLD=0).  A later re-tune keeps working, without the fold on those blocks.
        -> bytes released"""
        blocks = getattr(self, "_blocks", None)
        if blocks is None or not hasattr(blocks[0], "qkv_wf"):
            return 0
        enc = self.cfg.enc
        layers = len(blocks)
        bf16_stream = enc.residual_stream == 1
        fp8 = self.precision == "fp8"
        first = enc.fp8_first_layer if fp8 else layers
        first_mlp = first - (enc.fp8_mlp_extra if fp8 else 0)
        freed = 0
        for i in range(layers):
            b = blocks[i]
            for name, alive in (("qkv", bf16_stream and i < first), ("fc1", bf16_stream and i < first_mlp)):
                if alive:
                    continue
                for suffix in ("_wf", "_bf", "_sf"):
                    ptr = getattr(b, name + suffix)
                    if ptr:
                        freed += self._h.drop(ptr)
                        setattr(b, name + suffix, None)
        if freed:
            getattr(self, "_graphs", {}).clear()     # captured launch sequences may have baked the folded form in
            for _, q in (getattr(self, "_queues", None) or {}).values():    # ... and so may the native queues' (engine/native_queue.py)
                q.close()
            if getattr(self, "_queues", None):
                self._queues = {}
        return freed

    # towers whose output is ONE row per item (class token / EOT): the last block's out-proj and MLP run on those rows only (towers.hip,
    # last_block_selected — in bf16, whatever the policy says), so of that block only the QKV GEMM (3 of its 12 W^2) can be e4m3 work at all
    pools_one_row = False

    def _fp8_share(self, layers: int, split: int, extra: int) -> float:
        """e4m3 share of a tower's GEMM FLOPs, in blocks: blocks [split, layers) whole, the MLP halves (2/3) of blocks [split - extra, split)"""
        share = (layers - split) + (2.0 / 3.0) * extra
        if self.pools_one_row:
            if split < layers:
                share -= 0.75                       # the last block counts for its QKV GEMM only
            elif extra >= 1:
                share -= 2.0 / 3.0                  # ... and its MLP half for nothing
        return share

    def __init__(self, device: str):
        self.device = _require_gpu(device)
        self.lib = L.load()
        self._h = _Holder(self.device)
        # Concurrent callers (the reference serves 8 index + 8 search FastAPI threads, api/configs.py:27-28) do not serialise on the
        # tower: every calling thread owns its scratch workspace (bump-allocated per call, a few hundred MB against 288 GB of HBM) and
        # enqueues on ITS current stream, so one request's H2D / host work overlaps another's kernels.  Only the hipGraph cache
        # (static buffers shared by all callers of one shape) stays behind the lock.
        self._tls = threading.local()
        self._lock = threading.Lock()
        self._graphs: Dict[tuple, "_GraphedCall"] = {}
        self._graphs_off = False
        # The drop-in boundary as north_star words it: the towers' entry points are PyTorch-ROCm custom ops (torch.ops.marqo_hip.*,
        # csrc/torch_ops.cpp) that forward to the C ABI on PyTorch's current stream; MARQO_AMD_BOUNDARY=ctypes calls the C ABI directly
        # (what a non-torch host binds).  Same kernels, same arguments either way (tests/test_torch_ops_gpu.py: bit-identical).
        self._ops = L.load_torch_ops() if L.boundary() == "torch_ops" else None
        self._blobs = None

    def _forget_queue(self, q) -> None:
        """a native queue that was closed under a caller (engine/native_queue.gone): the next small call creates a fresh one"""
        with self._lock:
            for k, ent in list((getattr(self, "_queues", None) or {}).items()):
                if ent[1] is q:
                    del self._queues[k]

    def _desc(self):
        """(cfg, weights) descriptors as the CPU byte tensors the custom ops take — zero-copy aliases of the ctypes structs"""
        if self._blobs is None:
            self._blobs = (L.struct_blob(self.cfg), L.struct_blob(self.w))
        return self._blobs

    def _call_image(self, kind: str, pixels: Tensor, n: int, out: Tensor, normalize: bool, ws: Tensor) -> None:
        """mq_encode_image_u8 / _f32 on the current stream of this thread, through the selected boundary"""
        if self._ops is not None:
            cfg, w = self._desc()
            (self._ops.encode_image_u8 if kind == "u8" else self._ops.encode_image_f32)(cfg, w, pixels, out, bool(normalize), ws)
            return
        fn = self.lib.mq_encode_image_u8 if kind == "u8" else self.lib.mq_encode_image_f32
        L.check(fn(C.byref(self.cfg), C.byref(self.w), pixels.data_ptr(), n, out.data_ptr(), 1 if normalize else 0, ws.data_ptr(),
                   ws.numel(), self._stream()), "mq_encode_image")

    def _call_text(self, clip: bool, d_packed: Tensor, d_cu: Tensor, cu: Tensor, nseq: int, d_pool: Optional[Tensor], out: Tensor,
                   normalize: bool, ws: Tensor) -> None:
        """mq_encode_clip_text / mq_encode_bert on the current stream of this thread, through the selected boundary"""
        if self._ops is not None:
            cfg, w = self._desc()
            if clip:
                self._ops.encode_clip_text(cfg, w, d_packed, d_cu, cu, d_pool, out, bool(normalize), ws)
            else:
                self._ops.encode_bert(cfg, w, d_packed, d_cu, cu, out, bool(normalize), ws)
            return
        if clip:
            L.check(self.lib.mq_encode_clip_text(C.byref(self.cfg), C.byref(self.w), d_packed.data_ptr(), d_cu.data_ptr(), cu.data_ptr(), nseq,
                                                 L.ptr(d_pool), out.data_ptr(), 1 if normalize else 0, ws.data_ptr(), ws.numel(),
                                                 self._stream()), "mq_encode_clip_text")
        else:
            L.check(self.lib.mq_encode_bert(C.byref(self.cfg), C.byref(self.w), d_packed.data_ptr(), d_cu.data_ptr(), cu.data_ptr(), nseq,
                                            out.data_ptr(), 1 if normalize else 0, ws.data_ptr(), ws.numel(), self._stream()),
                    "mq_encode_bert")

    def _graphs_ok(self) -> bool:
        """single-request calls replay a captured hipGraph (MARQO_AMD_GRAPHS=0 turns that off); an fp8 tower only once its scales are frozen"""
        return GRAPHS and not self._graphs_off and (self._fp8 is None or self._fp8.calibrated)

    def _capture(self, key: tuple, make) -> Optional["_GraphedCall"]:
        """graph for `key`, captured on first use; a failed capture turns graph replay off for this tower (eager launches remain)"""
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= MAX_GRAPHS_PER_TOWER:
                return None
            try:
                g = self._graphs[key] = make()
            except RuntimeError as e:
                self._graphs_off = True
                import logging
                logging.getLogger(__name__).warning("hipGraph capture failed (%s); this tower keeps launching eagerly", e)
                return None
        return g

    def _workspace(self, nbytes: int) -> Tensor:
        """this thread's scratch buffer OF THE CURRENT STREAM.  A block is allocated while its stream is current and only ever used from that
        stream (the caching allocator then orders its reuse behind that stream's work); a thread that alternates between streams — the
        two-stream image pipeline of open_clip_model.encode_image — keeps one block per stream instead of re-allocating at every switch."""
        by_stream = getattr(self._tls, "ws", None)
        if by_stream is None:
            by_stream = self._tls.ws = {}
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = by_stream.get(key)
        if ws is None or ws.numel() < nbytes:
            if len(by_stream) >= 4:     # (streams a thread no longer uses: do not keep their scratch)
                by_stream.clear()
            ws = by_stream[key] = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return ws

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def weight_bytes(self) -> int:
        return self._h.nbytes()


class VitTower(_TowerBase):
    """CLIP ViT image tower (open_clip `visual.*` checkpoint tensors); arch.pool == "map": the timm SigLIP ViT behind open_clip's
    TimmModel (`visual.trunk.*`)."""

    def __init__(self, arch: VitArch, sd: Dict[str, Tensor], device: str,
                 mean: Sequence[float] = OPENAI_DATASET_MEAN, std: Sequence[float] = OPENAI_DATASET_STD,
                 precision: str = "bf16"):
        super().__init__(device)
        if precision not in ("bf16", "fp8"):
            raise ValueError(f"precision must be 'bf16' or 'fp8', got {precision!r}")
        self.precision = precision
        self.arch = arch
        W, P = arch.width, arch.patch_size
        if arch.image_size < P:
            raise ValueError("image_size must be at least one patch")
        # (an image size that is not a multiple of the patch — ViT-SO400M-14-SigLIP-384: 384 = 27 * 14 + 6 — follows the strided conv:
        # floor(S / P) patches per side, the trailing pixels are not read)
        K = 3 * P * P
        Kp = (K + 63) // 64 * 64
        h = self._h
        patch_w = torch.zeros(W, Kp, dtype=torch.float32)
        if arch.pool == "map":
            # timm SigLIP ViT as open_clip's `visual.trunk` (no class token, no ln_pre, conv bias, attention-pool head, no proj)
            t = "visual.trunk."
            F = arch.mlp_dim
            hd = _head_dim(W, arch.heads)
            if arch.out_dim != W:
                raise ValueError(f"SigLIP towers have no visual projection: out_dim {arch.out_dim} must equal the width {W}")
            patch_w[:, :K] = _need(sd, t + "patch_embed.proj.weight", (W, 3, P, P)).detach().to(torch.float32).reshape(W, K)
            pos = _need(sd, t + "pos_embed", (1, arch.tokens, W)).detach().to(torch.float32)[0] + \
                _need(sd, t + "patch_embed.proj.bias", (W,)).detach().to(torch.float32)  # the conv bias rides on the position table
            self._blocks = _clip_blocks(h, sd, t, arch.layers, W, F, arch.heads, keys=_TIMM_KEYS)
            a = t + "attn_pool."
            f32 = lambda k, shape: _need(sd, a + k, shape).detach().to(torch.float32)
            # the single learned query is a constant of the model: q = Linear_q(latent) / sqrt(head dim), in fp32 at load
            q = (f32("latent", (1, 1, W)).reshape(W) @ f32("q.weight", (W, W)).t() + f32("q.bias", (W,))) * hd ** -0.5
            m1_w, m1_b, m2_w = _pad_mlp(f32("mlp.fc1.weight", (F, W)), f32("mlp.fc1.bias", (F,)), f32("mlp.fc2.weight", (W, F)))
            self._map = L.MapHead(q=h.f32(q), kv_w=h.bf16(f32("kv.weight", (2 * W, W))), kv_b=h.f32(f32("kv.bias", (2 * W,))),
                                  proj_w=h.bf16(f32("proj.weight", (W, W))), proj_b=h.f32(f32("proj.bias", (W,))),
                                  ln_g=h.f32(f32("norm.weight", (W,))), ln_b=h.f32(f32("norm.bias", (W,))),
                                  fc1_w=h.bf16(m1_w), fc1_b=h.f32(m1_b), fc2_w=h.bf16(m2_w), fc2_b=h.f32(f32("mlp.fc2.bias", (W,))))
            self.w = L.VitWeights(patch_w=h.bf16(patch_w), cls=None, pos=h.f32(pos), ln_pre_g=None, ln_pre_b=None, blocks=self._blocks,
                                  ln_post_g=h.f32(_need(sd, t + "norm.weight", (W,))), ln_post_b=h.f32(_need(sd, t + "norm.bias", (W,))),
                                  proj_w=None, map=C.pointer(self._map))
            pool, map_mlp = L.MQ_VIT_POOL_MAP, _ceil64(F)
        elif arch.pool == "query":
            # open_clip VisionTransformer + AttentionalPooler (CoCa, model_registry.py:344-370): the CLIP trunk, then one learned query over
            # ln_k(tokens) in a MultiheadAttention of width D = out_dim with kdim = vdim = W, ln_post over D, proj [D, D]
            D, Hp = arch.out_dim, arch.pool_heads
            if D % Hp or (D // Hp) % 8 or D // Hp > 128 or D % 64 or D > W:
                raise ValueError(f"attentional pooler of width {D} with {Hp} heads is not runnable (head dim a multiple of 8, <= 128; width a multiple of 64, <= {W})")
            patch_w[:, :K] = _need(sd, "visual.conv1.weight", (W, 3, P, P)).detach().to(torch.float32).reshape(W, K)
            self._blocks = _clip_blocks(h, sd, "visual.transformer.", arch.layers, W, arch.mlp_dim, arch.heads)
            a = "visual.attn_pool."
            f32 = lambda k, shape: _need(sd, a + k, shape).detach().to(torch.float32)
            bias = f32("attn.in_proj_bias", (3 * D,))
            query = _need(sd, a + "query", None).detach().to(torch.float32)
            if query.ndim != 2 or query.shape[1] != D:
                raise ValueError(f"checkpoint tensor '{a}query' has shape {tuple(query.shape)}, expected [n_queries, {D}]")
            # only the FIRST learned query reaches the contrastive embedding (pooled = attn_pool(x)[:, 0]); it is a constant of the model:
            # q0 = (ln_q(query)[0] @ Wq^T + bq) / sqrt(head dim), in fp32 at load
            q0 = torch.nn.functional.layer_norm(query[:1], (D,), f32("ln_q.weight", (D,)), f32("ln_q.bias", (D,)), arch.ln_eps)[0]
            q0 = (q0 @ f32("attn.q_proj_weight", (D, D)).t() + bias[:D]) * (D // Hp) ** -0.5
            kv_w = torch.cat([f32("attn.k_proj_weight", (D, W)), f32("attn.v_proj_weight", (D, W))], dim=0)
            self._map = L.MapHead(q=h.f32(q0), kv_w=h.bf16(kv_w), kv_b=h.f32(bias[D:]),
                                  proj_w=h.bf16(f32("attn.out_proj.weight", (D, D))), proj_b=h.f32(f32("attn.out_proj.bias", (D,))),
                                  ln_g=h.f32(_need(sd, "visual.ln_post.weight", (D,))), ln_b=h.f32(_need(sd, "visual.ln_post.bias", (D,))),
                                  fc1_w=None, fc1_b=None, fc2_w=None, fc2_b=None)
            self.w = L.VitWeights(
                patch_w=h.bf16(patch_w), cls=h.f32(_need(sd, "visual.class_embedding", (W,))),
                pos=h.f32(_need(sd, "visual.positional_embedding", (arch.tokens, W))),
                ln_pre_g=h.f32(_need(sd, "visual.ln_pre.weight", (W,))), ln_pre_b=h.f32(_need(sd, "visual.ln_pre.bias", (W,))),
                blocks=self._blocks, ln_post_g=h.f32(f32("ln_k.weight", (W,))), ln_post_b=h.f32(f32("ln_k.bias", (W,))),   # the norm every token takes before k | v
                proj_w=h.bf16(_need(sd, "visual.proj", (D, D)).detach().to(torch.float32).t()), map=C.pointer(self._map))
            pool, map_mlp = L.MQ_VIT_POOL_QUERY, 0
        elif arch.eva:
            # timm Eva as open_clip's `visual.trunk` (EVA02-CLIP): class token + learned positions (the conv bias rides on the patch rows of the
            # position table), rotary table computed here (a non-persistent buffer of the checkpoint), norm(class token) -> head
            if precision != "bf16" or arch.pool != "cls":
                raise ValueError("EVA02 towers run on the bf16 path with the class-token head")
            t = "visual.trunk."
            f32 = lambda k, shape: _need(sd, t + k, shape).detach().to(torch.float32)
            patch_w[:, :K] = f32("patch_embed.proj.weight", (W, 3, P, P)).reshape(W, K)
            pos = f32("pos_embed", (1, arch.tokens, W))[0].clone()
            pos[1:] += f32("patch_embed.proj.bias", (W,))
            self._blocks = _eva_blocks(h, sd, t, arch.layers, W, arch.mlp_dim, arch.heads)
            self._rope = h.f32(arch.rope_table())
            self.w = L.VitWeights(patch_w=h.bf16(patch_w), cls=h.f32(f32("cls_token", (1, 1, W)).reshape(W)), pos=h.f32(pos), ln_pre_g=None, ln_pre_b=None,
                                  blocks=self._blocks, ln_post_g=h.f32(f32("norm.weight", (W,))), ln_post_b=h.f32(f32("norm.bias", (W,))),
                                  proj_w=h.bf16(f32("head.weight", (arch.out_dim, W))), map=None, proj_b=h.f32(f32("head.bias", (arch.out_dim,))))
            pool, map_mlp = L.MQ_VIT_POOL_CLS, 0
        else:
            patch_w[:, :K] = _need(sd, "visual.conv1.weight", (W, 3, P, P)).detach().to(torch.float32).reshape(W, K)
            self._blocks = _clip_blocks(h, sd, "visual.transformer.", arch.layers, W, arch.mlp_dim, arch.heads)
            if arch.pool == "avg" and arch.ln_pre:
                raise ValueError("pool 'avg' is built for the no_ln_pre / final_ln_after_pool form (CLIPA)")
            self.w = L.VitWeights(
                patch_w=h.bf16(patch_w),
                cls=h.f32(_need(sd, "visual.class_embedding", (W,))),
                pos=h.f32(_need(sd, "visual.positional_embedding", (arch.tokens, W))),
                ln_pre_g=h.f32(_need(sd, "visual.ln_pre.weight", (W,))) if arch.ln_pre else None,
                ln_pre_b=h.f32(_need(sd, "visual.ln_pre.bias", (W,))) if arch.ln_pre else None,
                blocks=self._blocks,
                ln_post_g=h.f32(_need(sd, "visual.ln_post.weight", (W,))), ln_post_b=h.f32(_need(sd, "visual.ln_post.bias", (W,))),
                proj_w=h.bf16(_need(sd, "visual.proj", (W, arch.out_dim)).detach().to(torch.float32).t()), map=None)
            pool, map_mlp = (L.MQ_VIT_POOL_AVG if arch.pool == "avg" else L.MQ_VIT_POOL_CLS), 0
            self.pools_one_row = arch.pool != "avg"
        self.cfg = L.VitCfg(enc=_encoder_cfg(W, arch.layers, arch.heads, arch.mlp_dim, arch.quick_gelu, False,
                                             L.MQ_MASK_NONE, arch.ln_eps),
                            image_size=arch.image_size, patch_size=P, out_dim=arch.out_dim,
                            mean=(C.c_float * 3)(*mean), std=(C.c_float * 3)(*std), pool=pool, map_mlp_dim=map_mlp,
                            pool_dim=arch.out_dim if arch.pool == "query" else 0, pool_heads=arch.pool_heads if arch.pool == "query" else 0)
        if arch.eva:
            enc = self.cfg.enc
            enc.mlp_glu, enc.act, enc.mlp_ln_dim = (2 if EVA_GLU_EPILOGUE else 1), L.MQ_ACT_SILU, arch.mlp_dim     # 2: fc1 rows interleaved 16 by 16 (_eva_blocks), the product in the GEMM's epilogue
            enc.d_rope_table, enc.rope_prefix = self._rope, 1
        self.max_images_per_call = max(1, MAX_ROWS_PER_CALL // arch.tokens)
        # (rounds 1-6 kept a knob here that split a call into k sub-batches on k HIP streams; measured on the round-6 kernels: 256 images 93.1 k ->
        # 68.9 k embeddings/s at k = 2, 62.5 k at k = 3 — persistent GEMM grids leave a second stream nothing to fill — profiles/r08a_*; removed)
        if precision == "fp8":
            self._enable_fp8(self._blocks, arch.layers, W, _ceil64(arch.mlp_dim))
        self.cfg.enc.residual_stream = 2
        if precision == "bf16":
            self.tune_residual_default()

    def tune_residual_default(self) -> str:
        u8 = self.calibration_images()
        return self.tune_residual_stream(lambda: self.encode_u8(u8))

    def _run(self, kind: str, pixels: Tensor, normalize: bool) -> Tensor:
        n = pixels.shape[0]
        if n == 1 and self._graphs_ok():
            with self._lock, torch.cuda.device(self.device):
                def make():
                    inp = torch.empty_like(pixels)
                    o = torch.empty(1, self.arch.out_dim, dtype=torch.float32, device=self.device)
                    ws = torch.empty(self.lib.mq_vit_workspace_bytes(C.byref(self.cfg), 1) + 256, dtype=torch.uint8, device=self.device)
                    inp.copy_(pixels)
                    return _GraphedCall(self.device, inp, o, (ws,), lambda: self._call_image(kind, inp, 1, o, normalize, ws))
                g = self._capture((pixels.dtype, bool(normalize)), make)
                if g is not None:
                    return g(pixels)
        out = torch.empty(n, self.arch.out_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device), _large_call(self.device, n * self.arch.tokens):
            for i in range(0, n, self.max_images_per_call):
                m = min(self.max_images_per_call, n - i)
                need = self.lib.mq_vit_workspace_bytes(C.byref(self.cfg), m)
                ws = self._workspace(need)
                self._call_image(kind, pixels[i:i + m], m, out[i:i + m], normalize, ws)
        return out

    # ---- native request queue (engine/native_queue.py, csrc/queue.hip): the request threads' small calls on preprocessed images share tower calls ----
    _queues: Optional[Dict[bool, tuple]] = None
    _queues_off = False

    def _queue(self, normalize: bool) -> Optional["NQ.ImageQueue"]:
        """this tower's image queue for `normalize` (created at first use, re-created when the tower's policy fields have changed), or None"""
        if not NQ.ENABLED or NQ.IMAGE_REQUEST_MAX <= 0 or self._queues_off or (self._fp8 is not None and not self._fp8.calibrated):
            return None
        sig = bytes(self.cfg)
        ent = (self._queues or {}).get(bool(normalize))
        if ent is not None and ent[0] == sig:
            return ent[1]
        with self._lock:
            if self._queues is None:
                self._queues = {}
            ent = self._queues.get(bool(normalize))
            if ent is not None and ent[0] == sig:
                return ent[1]
            if ent is not None:
                ent[1].close()
            try:
                idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
                q = NQ.ImageQueue(self.lib, self.cfg, self.w, idx, self.arch.out_dim, bool(normalize))
            except (L.MarqoHipUnavailableError, L.MarqoHipError) as e:
                self._queues_off = True
                import logging
                logging.getLogger(__name__).warning("native request queue unavailable (%s); small image calls keep the direct path", e)
                return None
            self._queues[bool(normalize)] = (sig, q)
            return q

    def queue_rows_images(self, tensors: Sequence[Tensor], normalize: bool = True) -> Optional[np.ndarray]:
        """the loaders' lean small-call path for preprocessed images: fp32 [3, S, S] device tensors (complete: the caller has seen their stream idle) ->
        host rows through the native queue, or None (no queue / not the queue's kind of call)"""
        n = len(tensors)
        if n < 1 or n > NQ.IMAGE_REQUEST_MAX:
            return None
        q = self._queue(normalize)
        if q is None:
            return None
        S = self.arch.image_size
        ptrs = []
        for t in tensors:
            if t.dtype != torch.float32 or t.device != self.device or tuple(t.shape) != (3, S, S) or not t.is_contiguous():
                return None
            ptrs.append(t.data_ptr())
        try:
            return q.encode_ptrs(ptrs)
        except L.MarqoHipError as e:
            if NQ.gone(e):
                self._forget_queue(q)
                return None
            raise

    def queue_stats(self) -> Dict[bool, Dict[str, int]]:
        return {k: ent[1].stats() for k, ent in (self._queues or {}).items()}

    def calibration_images(self, n: int = 16, seed: int = 0) -> Tensor:
        """the fixed, seeded calibration batch of the fp8 policy: half uniform pixel noise, half smooth low-frequency fields plus
        mild noise (closer to the spectrum of photographs) — uint8 [n, S, S, 3] on the device.  The same for every load of a
        checkpoint, so scales and block split are reproducible across restarts and replicas."""
        S = self.arch.image_size
        g = torch.Generator().manual_seed(1000 + seed)
        noise = torch.randint(0, 256, (n - n // 2, S, S, 3), generator=g, dtype=torch.uint8)
        coarse = torch.rand(n // 2, 3, 8, 8, generator=g)
        smooth = torch.nn.functional.interpolate(coarse, size=(S, S), mode="bilinear", align_corners=False)
        smooth = (smooth * 255 + 12 * torch.randn(smooth.shape, generator=g)).clamp(0, 255).permute(0, 2, 3, 1).to(torch.uint8)
        return torch.cat([noise, smooth], dim=0).contiguous().to(self.device)

    def tune_fp8_default(self, budget: Optional[float] = None) -> int:
        u8 = self.calibration_images()
        return self.tune_fp8(lambda: self.encode_u8(u8), budget=budget)

    def encode_u8(self, images_u8: Tensor, normalize: bool = True) -> Tensor:
        """uint8 [n, S, S, 3] (HWC RGB, on this device) -> fp32 [n, D] on device (async on the current stream)."""
        S = self.arch.image_size
        if images_u8.dtype != torch.uint8 or images_u8.ndim != 4 or tuple(images_u8.shape[1:]) != (S, S, 3):
            raise ValueError(f"expected uint8 [n, {S}, {S}, 3], got {images_u8.dtype} {tuple(images_u8.shape)}")
        images_u8 = images_u8.to(self.device, non_blocking=True).contiguous()
        return self._run("u8", images_u8, normalize)

    def encode_f32(self, pixels: Tensor, normalize: bool = True) -> Tensor:
        """preprocessed fp32 [n, 3, S, S] -> fp32 [n, D] on device."""
        S = self.arch.image_size
        if pixels.ndim != 4 or tuple(pixels.shape[1:]) != (3, S, S):
            raise ValueError(f"expected float [n, 3, {S}, {S}], got {tuple(pixels.shape)}")
        pixels = pixels.to(device=self.device, dtype=torch.float32, non_blocking=True).contiguous()
        return self._run("f32", pixels, normalize)


def _host_i64(t) -> np.ndarray:
    """ids / lengths / masks (torch tensor on any device, or array-like) -> int64 ndarray on the host.  The request path does its integer
    bookkeeping in NumPy on purpose: PyTorch's CPU kernels for boolean-mask indexing, reductions and index_select enter an OpenMP region
    with torch.get_num_threads() workers (128 on the GPU boxes) that keep spinning after a 10 k-element job; under a container CPU quota
    that spin throttled the whole process — 128 strings took 20-28 ms to pack instead of 0.07 ms (profiles/r02ae_ingest_phases.txt)."""
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=np.int64)


def _pack(ids: np.ndarray, lengths: np.ndarray) -> Tuple[Tensor, Tensor]:
    """right-padded [n, S] ids + lengths (host ndarrays) -> (packed int32 ids [rows], cu_seqlens int32 [n+1]) as host tensors."""
    n, S = ids.shape
    keep = np.arange(S)[None, :] < lengths[:, None]
    packed = np.ascontiguousarray(ids[keep], dtype=np.int32)
    cu = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(lengths, out=cu[1:])
    return torch.from_numpy(packed), torch.from_numpy(cu)


class _TextTowerBase(_TowerBase):
    # rows (tokens) per engine call.  Towers may lower it: the post-LN BERT encoders (fp32 stream: ~10 KB of activations per row and layer) run
    # 4-6 % faster per item at 20-40 k rows per call than at 79 k (profiles/r03v_batch_sweep.txt); the pre-LN towers do not (ViT-L/14: larger is better)
    max_rows_per_call = MAX_ROWS_PER_CALL

    def _chunks(self, lengths: np.ndarray):
        """Yield (start, stop) sequence ranges with <= max_rows_per_call rows each, of about EQUAL row counts (a greedy fill would leave a small,
        badly tiled last call).  Vectorised: the common case of one chunk costs one cumsum."""
        n = int(lengths.size)
        if n == 0:
            return
        cum = np.cumsum(lengths)
        total, limit = int(cum[-1]), int(self.max_rows_per_call)
        if total <= limit:
            yield 0, n
            return
        pieces = -(-total // limit)
        target = -(-total // pieces)
        start, base = 0, 0
        while start < n:
            stop = int(np.searchsorted(cum, base + target, side="right"))
            stop = max(stop, start + 1)
            if int(cum[stop - 1]) - base > limit and stop - 1 > start:   # (never over the hard limit)
                stop -= 1
            yield start, stop
            base = int(cum[stop - 1])
            start = stop

    def _to_device(self, t: Tensor) -> Tensor:
        """small host -> device copy through pinned memory (a pageable copy would synchronise the stream)"""
        return t.pin_memory().to(self.device, non_blocking=True)

    # ---- native request queue (engine/native_queue.py, csrc/queue.hip): concurrent small calls of the request threads share tower calls ----------
    _queues: Optional[Dict[bool, tuple]] = None
    _queues_off = False
    _active = 0                       # request-thread calls of the small-call path in flight on this tower
    _active_lock = threading.Lock()   # (class-wide: two increments)

    def _queue(self, normalize: bool, clip: bool) -> Optional["NQ.TextQueue"]:
        """this tower's queue for `normalize` (created at first use, re-created when the tower's policy fields have changed since: its scratch is
        sized from them), or None — switched off, or it could not be created (logged once; the direct path stays)"""
        if not NQ.ENABLED or self._queues_off or (self._fp8 is not None and not self._fp8.calibrated):
            return None
        sig = bytes(self.cfg)
        ent = (self._queues or {}).get(bool(normalize))
        if ent is not None and ent[0] == sig:
            return ent[1]
        with self._lock:
            if self._queues is None:
                self._queues = {}
            ent = self._queues.get(bool(normalize))
            if ent is not None and ent[0] == sig:
                return ent[1]
            if ent is not None:
                ent[1].close()
            try:
                idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
                max_len = (self.arch.ctx + (1 if getattr(self.arch, "cls_embed", False) else 0)) if clip else self.arch.max_pos
                q = NQ.TextQueue(self.lib, L.QUEUE_CLIP_TEXT if clip else L.QUEUE_BERT, self.cfg, self.w, idx,
                                 self.arch.out_dim if clip else self.out_width, max_len, bool(normalize))
            except (L.MarqoHipUnavailableError, L.MarqoHipError) as e:
                self._queues_off = True
                import logging
                logging.getLogger(__name__).warning("native request queue unavailable (%s); small text calls keep the direct path", e)
                return None
            self._queues[bool(normalize)] = (sig, q)
            return q

    def _small_call(self, ids_h: np.ndarray, lengths: np.ndarray, normalize: bool, clip: bool) -> Optional[Tensor]:
        """the request threads' small calls (loaders inside `request_stream`, host rows wanted): a LONE single-sequence call replays the captured
        launch sequence of its token count (lowest latency: no worker hand-over, ~100 launches as one graph); calls that find others in flight
        go through the native queue and share tower calls with them.  None: not a small call / neither is available — the caller launches eagerly."""
        n = int(lengths.size)
        if not getattr(_request_tls, "host_output", False) or n < 1 or n > NQ.MAX_SEQS:
            return None
        with _TextTowerBase._active_lock:
            self._active += 1
            alone = self._active == 1
        try:
            if n == 1 and alone and self._graphs_ok():
                one = self._encode_one(torch.from_numpy(ids_h[0, :int(lengths[0])]), normalize, clip)
                if one is not None:
                    return one
            q = self._queue(normalize, clip)
            if q is None or not q.takes(n, int(lengths.sum())):
                return None
            packed, _ = _pack(ids_h, lengths)
            try:
                return torch.from_numpy(q.encode(packed.numpy(), lengths))
            except L.MarqoHipError as e:
                if NQ.gone(e):
                    self._forget_queue(q)
                    return None
                raise
        finally:
            with _TextTowerBase._active_lock:
                self._active -= 1

    def queue_rows(self, ids_h: np.ndarray, lengths: np.ndarray, normalize: bool, clip: bool) -> Optional[np.ndarray]:
        """the loaders' LEAN small-call path — host ids in, host rows out, no stream context, no torch call, a handful of NumPy calls (with 16
        request threads in the interpreter every statement here is time the others wait for): this request's rows through the native queue (a lone
        single query included: the worker replays a hipGraph of its token count), or None when the call belongs to the regular path (no queue; too
        many sequences)"""
        n = int(lengths.size)
        if n < 1 or n > NQ.MAX_SEQS:
            return None
        with _TextTowerBase._active_lock:
            self._active += 1
            alone = self._active == 1
        try:
            if n == 1 and alone and not NQ.GRAPHS and self._graphs_ok():
                return None                       # (MARQO_AMD_NATIVE_QUEUE_GRAPHS=0: the lone query replays the tower's own captured graph, through torch)
            q = self._queue(normalize, clip)
            if q is None or not q.takes(n, int(lengths.sum())):
                return None
            packed = ids_h[0, :int(lengths[0])] if n == 1 else ids_h[np.arange(ids_h.shape[1])[None, :] < lengths[:, None]]
            try:
                return q.encode_raw(np.ascontiguousarray(packed, dtype=np.int32), np.ascontiguousarray(lengths, dtype=np.int32), n)
            except L.MarqoHipError as e:
                if NQ.gone(e):
                    self._forget_queue(q)
                    return None
                raise
        finally:
            with _TextTowerBase._active_lock:
                self._active -= 1

    def queue_stats(self) -> Dict[bool, Dict[str, int]]:
        return {k: ent[1].stats() for k, ent in (self._queues or {}).items()}

    def _encode_one(self, src_ids: Tensor, normalize: bool, clip: bool) -> Optional[Tensor]:
        """ONE sequence (1-D ids of its real length, host or device) through the captured launch sequence of that token count
        (None: capture is not available, the caller launches eagerly)"""
        n_tok = int(src_ids.numel())
        with self._lock, torch.cuda.device(self.device):
            def make():
                d_packed = torch.empty(n_tok, dtype=torch.int32, device=self.device)
                cu = torch.tensor([0, n_tok], dtype=torch.int32)
                d_cu = cu.to(self.device)
                o = torch.empty(1, self.arch.out_dim if clip else self.out_width, dtype=torch.float32, device=self.device)
                keep, d_pool = [], None
                if clip:
                    ws = torch.empty(self.lib.mq_clip_text_workspace_bytes(C.byref(self.cfg), n_tok, 1) + 256, dtype=torch.uint8, device=self.device)
                    d_pool = torch.tensor([n_tok - 1], dtype=torch.int32).to(self.device)   # the pooled row (last = EOT) is known at capture
                    keep.append(d_pool)
                else:
                    ws = torch.empty(self.lib.mq_bert_workspace_bytes(C.byref(self.cfg), n_tok, 1) + 256, dtype=torch.uint8, device=self.device)
                launch = lambda: self._call_text(clip, d_packed, d_cu, cu, 1, d_pool, o, normalize, ws)
                d_packed.copy_(src_ids)
                return _GraphedCall(self.device, d_packed, o, (cu, d_cu, ws, *keep), launch)
            g = self._capture((n_tok, bool(normalize)), make)
            if g is None:
                return None
            return g(src_ids.to(torch.int32) if src_ids.dtype != torch.int32 else src_ids)

    def _encode_device(self, d_ids: Tensor, lengths: Tensor, max_len: int, normalize: bool, clip: bool) -> Tensor:
        """device-resident right-padded ids + host lengths -> embeddings; packing to the towers' layout on the GPU"""
        if d_ids.ndim != 2 or d_ids.dtype != torch.int32 or d_ids.device != self.device:
            raise ValueError(f"expected int32 [n, S] ids on {self.device}, got {d_ids.dtype} {tuple(d_ids.shape)} on {d_ids.device}")
        d_ids = d_ids.contiguous()
        n, S = d_ids.shape
        lengths = _host_i64(lengths)
        if lengths.shape != (n,) or (n and (int(lengths.min()) < 1 or int(lengths.max()) > min(S, max_len))):
            raise ValueError(f"lengths must be [n] within [1, {min(S, max_len)}]")
        if n == 1 and self._graphs_ok():
            one = self._encode_one(d_ids[0, :int(lengths[0])], normalize, clip)
            if one is not None:
                return one
        out_dim = self.arch.out_dim if clip else self.out_width
        out = torch.empty(n, out_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device), _large_call(self.device, int(lengths.sum())):
            for a, b in self._chunks(lengths):
                cu_np = np.zeros(b - a + 1, dtype=np.int32)
                np.cumsum(lengths[a:b], out=cu_np[1:])
                cu = torch.from_numpy(cu_np)
                rows, nseq = int(cu_np[-1]), b - a
                d_cu = self._to_device(cu)
                d_packed = torch.empty(rows, dtype=torch.int32, device=self.device)
                L.check(self.lib.mq_pack_ids(d_ids[a:b].data_ptr(), S, d_cu.data_ptr(), nseq, d_packed.data_ptr(), self._stream()), "mq_pack_ids")
                need = (self.lib.mq_clip_text_workspace_bytes if clip else self.lib.mq_bert_workspace_bytes)(C.byref(self.cfg), rows, nseq)
                self._call_text(clip, d_packed, d_cu, cu, nseq, None, out[a:b], normalize, self._workspace(need))
        return out


class ClipTextTower(_TextTowerBase):
    """CLIP text tower (open_clip `token_embedding`, `transformer.*`, `ln_final`, `text_projection`)."""

    def __init__(self, arch: ClipTextArch, sd: Dict[str, Tensor], device: str, precision: str = "bf16"):
        super().__init__(device)
        if precision not in ("bf16", "fp8"):
            raise ValueError(f"precision must be 'bf16' or 'fp8', got {precision!r}")
        self.precision = precision
        self.arch = arch
        self.pools_one_row = True   # EOT (causal) / last (SigLIP) position
        W = arch.width
        h = self._h
        px = arch.prefix  # "" (CLIP) / "text." (SigLIP under open_clip's CustomTextCLIP)
        self._blocks = _clip_blocks(h, sd, px + "transformer.", arch.layers, W, arch.mlp_dim, arch.heads)
        if arch.proj_bias:  # SigLIP: text_projection is a Linear (weight [D, W] + bias)
            proj_w = _need(sd, px + "text_projection.weight", (arch.out_dim, W)).detach().to(torch.float32)
            proj_b = h.f32(_need(sd, px + "text_projection.bias", (arch.out_dim,)))
        else:
            proj_w = _need(sd, px + "text_projection", (W, arch.out_dim)).detach().to(torch.float32).t()
            proj_b = None
        tok_emb = _need(sd, px + "token_embedding.weight", (arch.vocab, W))
        if arch.cls_embed:
            # CoCa: the learned class embedding rides as two more rows of the token table (ids vocab, vocab + 1: the same vector); encode_ids packs
            # [text, ONE pad row, class] — or [text, class, class] for a text that fills all ctx - 1 positions — and the embedding kernel gives a
            # sequence's last row position ctx - 1 (cfg.cls_pos).  Why a pad row: open_clip 2.24.0's build_cls_mask pads the class row's key mask on
            # the LEFT, so the class token sees the text, the FIRST pad position and NOT itself (itself only behind a full-length text: the twin row
            # has the class token's K / V) — MQ_MASK_CAUSAL_CLS (marqo_hip.h); the pretrained coca_* checkpoints were trained with that mask.
            cls = _need(sd, px + "cls_emb", (W,)).detach().to(torch.float32)[None]
            tok_emb = torch.cat([tok_emb.detach().to(torch.float32), cls, cls], dim=0)
        self.w = L.ClipTextWeights(
            tok_emb=h.f32(tok_emb),
            pos=h.f32(_need(sd, px + "positional_embedding", (arch.ctx, W))),
            blocks=self._blocks,
            ln_final_g=h.f32(_need(sd, px + "ln_final.weight", (W,))), ln_final_b=h.f32(_need(sd, px + "ln_final.bias", (W,))),
            proj_w=h.bf16(proj_w), proj_b=proj_b)
        self.cfg = L.ClipTextCfg(enc=_encoder_cfg(W, arch.layers, arch.heads, arch.mlp_dim, arch.quick_gelu, False,
                                                  (L.MQ_MASK_CAUSAL_CLS if arch.cls_embed else L.MQ_MASK_CAUSAL) if arch.causal else L.MQ_MASK_NONE, arch.ln_eps),
                                 vocab=arch.vocab + (2 if arch.cls_embed else 0), ctx=arch.ctx, out_dim=arch.out_dim,
                                 cls_pos=arch.ctx - 1 if arch.cls_embed else 0)
        if arch.cls_embed and not arch.causal:
            raise ValueError("a class-embedding text tower is causal (CoCa)")
        if arch.cls_embed and precision == "fp8":
            raise ValueError("a class-embedding text tower (CoCa) runs on bf16 operands (its attention mask has no e4m3-output kernel)")
        if precision == "fp8":
            self._enable_fp8(self._blocks, arch.layers, W, _ceil64(arch.mlp_dim))
        self.cfg.enc.residual_stream = 2
        if precision == "bf16":
            self.tune_residual_default()

    def _with_cls(self, ids_h: np.ndarray) -> np.ndarray:
        """CoCa: [n, S <= ctx - 1] token ids (SOT ... EOT 0 ...) -> [n, ctx + 1] rows  text, 0 (ONE pad position), class id  — or, for a text that
        fills all ctx - 1 positions,  text, class id, class twin id  (see __init__: the class token attends the first pad position and not
        itself; behind a full text it attends itself, which the twin row stands for).  The last id of a row is its largest, so the EOT logic
        downstream (argmax = pooled position, rows up to it) lands on it."""
        n, S = ids_h.shape
        full = self.arch.ctx - 1
        if S > full:
            raise ValueError(f"a class-embedding text tower takes at most {full} token positions, got {S}")
        out = np.zeros((n, full + 2), dtype=np.int64)
        out[:, :S] = ids_h
        L_ = ids_h.argmax(axis=1) + 1                    # text length SOT .. EOT
        r = np.arange(n)
        is_full = L_ >= full
        out[r[~is_full], L_[~is_full] + 1] = self.arch.vocab          # text, pad (0), class
        out[r[is_full], full] = self.arch.vocab                       # text, class, twin
        out[r[is_full], full + 1] = self.arch.vocab + 1
        return out

    def tune_residual_default(self) -> str:
        ids = self.calibration_ids()
        return self.tune_residual_stream(lambda: self.encode_ids(ids))

    def calibration_ids(self, n: int = 32, seed: int = 0) -> Tensor:
        """fixed, seeded calibration texts of the fp8 policy: int64 [n, ctx] rows SOT, L random ids, EOT, zero padding with L spread
        over 3 .. ctx - 2 (SigLIP towers: random ids then pad-id padding, all ctx positions run)"""
        a = self.arch
        g = torch.Generator().manual_seed(2000 + seed)
        if not a.causal:
            ids = torch.full((n, a.ctx), a.pad_id, dtype=torch.int64)
            for i in range(n):
                L_ = 3 + (i * (a.ctx - 4)) // max(n - 1, 1)
                ids[i, :L_] = torch.randint(2, a.vocab, (L_,), generator=g)
            return ids
        S = a.ctx - 1 if a.cls_embed else a.ctx
        ids = torch.zeros(n, S, dtype=torch.int64)
        for i in range(n):
            L_ = 3 + (i * (S - 5)) // max(n - 1, 1)
            ids[i, 0] = a.vocab - 2
            ids[i, 1:1 + L_] = torch.randint(1, a.vocab - 2, (L_,), generator=g)
            ids[i, 1 + L_] = a.vocab - 1
        return ids

    def tune_fp8_default(self, budget: Optional[float] = None) -> int:
        ids = self.calibration_ids()
        return self.tune_fp8(lambda: self.encode_ids(ids), budget=budget)

    def encode_ids(self, ids: Tensor, normalize: bool = True, pack: bool = True) -> Tensor:
        """ids: int [n, ctx] zero-padded CLIP token ids (SOT ... EOT 0 0 ...), host or device.
        pack=True runs each sequence only up to its EOT (= argmax id, the pooled position): the later
        positions cannot influence that row under the causal mask.  pack=False runs all ctx positions
        like the reference does."""
        if ids.ndim != 2 or ids.shape[1] > self.arch.ctx:
            raise ValueError(f"expected ids [n, <= {self.arch.ctx}], got {tuple(ids.shape)}")
        ids_h = _host_i64(ids)
        if self.arch.cls_embed:
            ids_h, pack = self._with_cls(ids_h), True     # (the un-packed form would run the padding between the text and the class token)
        n, S = ids_h.shape
        if not self.arch.causal:
            # SigLIP: no mask at all — every one of the ctx positions (padding included) is attended to and the pooled row is the
            # last one, so all S = ctx positions run and there is nothing to pack away
            if S != self.arch.ctx:
                raise ValueError(f"an unmasked text tower runs exactly ctx = {self.arch.ctx} positions per text, got {S}")
            pack = False
        eot = np.full(n, S - 1, dtype=np.int64) if not self.arch.causal else ids_h.argmax(axis=1)
        lengths = (eot + 1) if pack else np.full(n, S, dtype=np.int64)
        if pack or not self.arch.causal:          # (the pooled row is each sequence's last: what the queue and the captured graphs pool)
            small = self._small_call(ids_h, lengths, normalize, clip=True)
            if small is not None:
                return small
        if n == 1 and (pack or not self.arch.causal) and self._graphs_ok():
            one = self._encode_one(torch.from_numpy(ids_h[0, :int(lengths[0])]), normalize, clip=True)
            if one is not None:
                return one
        out = torch.empty(n, self.arch.out_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device), _large_call(self.device, int(lengths.sum())):
            for a, b in self._chunks(lengths):
                packed, cu = _pack(ids_h[a:b], lengths[a:b])
                pool_rows = None if pack else torch.from_numpy((cu.numpy()[:-1].astype(np.int64) + eot[a:b]).astype(np.int32))
                d_ids = self._to_device(packed)
                d_cu = self._to_device(cu)
                d_pool = self._to_device(pool_rows) if pool_rows is not None else None
                rows, nseq = packed.numel(), b - a
                need = self.lib.mq_clip_text_workspace_bytes(C.byref(self.cfg), rows, nseq)
                self._call_text(True, d_ids, d_cu, cu, nseq, d_pool, out[a:b], normalize, self._workspace(need))
        return out


    def queue_rows_ids(self, ids_h: np.ndarray, normalize: bool = True) -> Optional[np.ndarray]:
        """`queue_rows` for the tokeniser's [n, <= ctx] id matrix (SOT ... EOT 0 ...): each sequence up to its EOT, as encode_ids(pack=True) runs it.
        None for the towers whose rows are not 'text up to EOT' (unmasked SigLIP towers, CoCa's appended class token): they keep the regular path."""
        if not self.arch.causal or self.arch.cls_embed or ids_h.ndim != 2 or ids_h.shape[1] > self.arch.ctx:
            return None
        return self.queue_rows(ids_h, ids_h.argmax(axis=1) + 1, normalize, clip=True)

    def encode_device(self, d_ids: Tensor, lengths: Tensor, normalize: bool = True) -> Tensor:
        """ids already on the device (engine/gpu_tokenizers.py): int32 [n, S] rows SOT ... EOT 0 ..., lengths int64 [n] on the
        host = SOT..EOT length.  Same as encode_ids(pack=True), but the packing runs on the GPU (mq_pack_ids)."""
        if not self.arch.causal and (d_ids.shape[1] != self.arch.ctx or int(_host_i64(lengths).min(initial=self.arch.ctx)) != self.arch.ctx):
            raise ValueError(f"an unmasked text tower runs exactly ctx = {self.arch.ctx} positions per text")
        if self.arch.cls_embed:   # the class id is appended on the host (a [n, S] id matrix: KBs)
            ids_h = _host_i64(d_ids)
            return self.encode_ids(torch.from_numpy(np.where(np.arange(ids_h.shape[1])[None] < _host_i64(lengths)[:, None], ids_h, 0)), normalize=normalize)
        return self._encode_device(d_ids, lengths, self.arch.ctx, normalize, clip=True)


class BertTower(_TextTowerBase):
    """BERT-family encoder + pooling (HF `BertModel` checkpoint tensors, with or without a `bert.` prefix)."""

    max_rows_per_call = _env_int("MARQO_AMD_BERT_ROWS_PER_CALL", 40 * 1024, lo=1024)   # see _TextTowerBase.max_rows_per_call

    def __init__(self, arch: BertArch, sd: Dict[str, Tensor], device: str, pooling: str = "mean", precision: str = "bf16"):
        super().__init__(device)
        if precision not in ("bf16", "fp8"):
            raise ValueError(f"precision must be 'bf16' or 'fp8', got {precision!r}")
        self.precision = precision
        self.arch = arch
        if pooling not in ("mean", "cls"):
            raise ValueError(f"pooling must be 'mean' or 'cls', got {pooling!r}")
        self.pooling = pooling
        for prefix in ("bert.", "roberta.", "new.", "mpnet."):  # HF checkpoints may carry the task-model prefix
            if "embeddings.word_embeddings.weight" not in sd and prefix + "embeddings.word_embeddings.weight" in sd:
                sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        W, F = arch.width, arch.mlp_dim
        hd = _head_dim(W, arch.heads)
        h = self._h
        arr = (L.BlockWeights * arch.layers)()
        new_model = arch.glu or arch.rope_theta is not None   # Alibaba-NLP NewModel naming (stella_en_400M_v5, gte-*-en-v1.5)
        if new_model and (precision != "bf16" or hd != _kernel_head_dim(hd, arch.heads)):
            raise ValueError("NewModel (rotary / gated-MLP) encoders run on the bf16 path with 64-wide heads")
        mpnet = arch.rel_buckets > 0   # MPNetModel naming: attention.attn.{q,k,v,o}, attention.LayerNorm, + one relative-position bias table
        if mpnet and (precision != "bf16" or hd != 64):
            raise ValueError("MPNet encoders (relative-position attention bias) run on the bf16 path with 64-wide heads")
        # checkpoint key names of the attention sub-block: (q, k, v, out-projection, LayerNorm)
        ak = ("attention.attn.q", "attention.attn.k", "attention.attn.v", "attention.attn.o", "attention.LayerNorm") if mpnet else \
             ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense", "attention.output.LayerNorm")
        for i in range(arch.layers):
            p = f"encoder.layer.{i}."
            b = arr[i]
            if new_model:
                qkv_w = _need(sd, p + "attention.qkv_proj.weight", (3 * W, W)).detach().float()
                qkv_b = _need(sd, p + "attention.qkv_proj.bias", (3 * W,)).detach().float()
                b.qkv_w, b.qkv_b = h.bf16(qkv_w), h.f32(qkv_b)
                b.out_w = h.bf16(_need(sd, p + "attention.o_proj.weight", (W, W)))
                b.out_b = h.f32(_need(sd, p + "attention.o_proj.bias", (W,)))
                b.ln1_g, b.ln1_b = h.f32(_need(sd, p + "attn_ln.weight", (W,))), h.f32(_need(sd, p + "attn_ln.bias", (W,)))
                b.fc1_w = h.bf16(_need(sd, p + "mlp.up_gate_proj.weight", (2 * F, W)))      # rows [0, F) = up, [F, 2F) = gate
                b.fc1_b = h.f32(sd[p + "mlp.up_gate_proj.bias"]) if p + "mlp.up_gate_proj.bias" in sd else None
                b.fc2_w = h.bf16(_need(sd, p + "mlp.down_proj.weight", (W, F)))
                b.fc2_b = h.f32(_need(sd, p + "mlp.down_proj.bias", (W,)))
                b.ln2_g, b.ln2_b = h.f32(_need(sd, p + "mlp_ln.weight", (W,))), h.f32(_need(sd, p + "mlp_ln.bias", (W,)))
                continue
            qkv_w = torch.cat([_need(sd, p + f"{n}.weight", (W, W)).detach().float() for n in ak[:3]], 0)
            qkv_b = torch.cat([_need(sd, p + f"{n}.bias", (W,)).detach().float() for n in ak[:3]], 0)
            out_w = _need(sd, p + ak[3] + ".weight", (W, W)).detach().float()
            if hd != _kernel_head_dim(hd, arch.heads):  # e5-small / bge-small / MiniLM: 12 heads of 32
                qkv_w, qkv_b, out_w = _pad_heads(qkv_w, qkv_b, out_w, arch.heads, hd)
            b.qkv_w, b.qkv_b = h.bf16(qkv_w), h.f32(qkv_b)
            b.out_w = h.bf16(out_w)
            b.out_b = h.f32(_need(sd, p + ak[3] + ".bias", (W,)))
            b.ln1_g = h.f32(_need(sd, p + ak[4] + ".weight", (W,)))
            b.ln1_b = h.f32(_need(sd, p + ak[4] + ".bias", (W,)))
            b.fc1_w = h.bf16(_need(sd, p + "intermediate.dense.weight", (F, W)))
            b.fc1_b = h.f32(_need(sd, p + "intermediate.dense.bias", (F,)))
            b.fc2_w = h.bf16(_need(sd, p + "output.dense.weight", (W, F)))
            b.fc2_b = h.f32(_need(sd, p + "output.dense.bias", (W,)))
            b.ln2_g = h.f32(_need(sd, p + "output.LayerNorm.weight", (W,)))
            b.ln2_b = h.f32(_need(sd, p + "output.LayerNorm.bias", (W,)))
        self._blocks = arr
        self.w = L.BertWeights(
            word_emb=h.f32(_need(sd, "embeddings.word_embeddings.weight", (arch.vocab, W))),
            # XLM-RoBERTa: position ids start at pos_offset -> hand the library the table from that row on; rotary models have no table
            pos_emb=None if arch.rope_theta is not None else
            h.f32(_need(sd, "embeddings.position_embeddings.weight", (arch.max_pos + arch.pos_offset, W))[arch.pos_offset:]),
            type_emb=h.f32(sd["embeddings.token_type_embeddings.weight"]) if "embeddings.token_type_embeddings.weight" in sd or not (new_model or mpnet)
            else None,
            emb_ln_g=h.f32(_need(sd, "embeddings.LayerNorm.weight", (W,))),
            emb_ln_b=h.f32(_need(sd, "embeddings.LayerNorm.bias", (W,))),
            blocks=arr)
        self.cfg = L.BertCfg(enc=_encoder_cfg(W, arch.layers, arch.heads, F, False, True, L.MQ_MASK_NONE, arch.ln_eps),
                             vocab=arch.vocab, max_pos=arch.max_pos,
                             pool=L.MQ_POOL_MEAN if pooling == "mean" else L.MQ_POOL_CLS)
        if mpnet:
            self._rel_bias = arch.rel_bias_table(_need(sd, "encoder.relative_attention_bias.weight", (arch.rel_buckets, arch.heads))).to(self.device)
            self.cfg.enc.d_rel_bias = self._rel_bias.data_ptr()
            self.cfg.enc.rel_span = arch.max_pos
        if arch.glu:
            self.cfg.enc.mlp_glu = 1
        if arch.rope_theta is not None:
            self._rope = arch.rope_inv_freq().to(self.device).contiguous()
            self.cfg.enc.d_rope_inv_freq = self._rope.data_ptr()
        if precision == "fp8":
            self._enable_fp8(self._blocks, arch.layers, W, F)
        self.cfg.enc.residual_stream = 2
        if precision == "bf16" and type(self) is BertTower:      # (towers with a projection head tune once the head is in place)
            self.tune_residual_default()

    def tune_residual_default(self) -> str:
        """post-LN form of the bf16 stream (towers.hip, stream_post16): the normalised bf16 rows are the residual; decided like the pre-LN form,
        on the fixed calibration batch, against the fp32-stream run of the same tower"""
        if self.arch.glu or self.arch.rope_theta is not None:
            self.cfg.enc.residual_stream, self.residual_stream = 2, "fp32"
            return self.residual_stream
        ids, mask = self.calibration_batch()
        return self.tune_residual_stream(lambda: self.encode_ids(ids, mask))

    @property
    def out_width(self) -> int:
        """columns of the embeddings: the encoder width, or the projection head's out_dim (HfClipTextTower)"""
        return int(self.cfg.out_dim) if self.cfg.out_dim else self.arch.width

    def calibration_batch(self, n: int = 32, seed: int = 0) -> Tuple[Tensor, Tensor]:
        """fixed, seeded calibration texts of the fp8 policy: (ids, attention_mask) int64 [n, S], lengths spread over 4 .. min(128, max_pos)"""
        a = self.arch
        g = torch.Generator().manual_seed(3000 + seed)
        S = min(128, a.max_pos)
        ids = torch.zeros(n, S, dtype=torch.int64)
        mask = torch.zeros(n, S, dtype=torch.int64)
        lo = min(1000, a.vocab // 2)
        for i in range(n):
            L_ = 4 + (i * (S - 4)) // max(n - 1, 1)
            ids[i, :L_] = torch.randint(lo, a.vocab, (L_,), generator=g)
            ids[i, 0], ids[i, L_ - 1] = (101, 102) if a.vocab > 102 and a.pos_offset == 0 else (0, 2)
            mask[i, :L_] = 1
        return ids, mask

    def tune_fp8_default(self, budget: Optional[float] = None) -> int:
        ids, mask = self.calibration_batch()
        return self.tune_fp8(lambda: self.encode_ids(ids, mask), budget=budget)

    def encode_ids(self, ids: Tensor, attention_mask: Tensor, normalize: bool = True) -> Tensor:
        """ids / attention_mask: int [n, S] as produced by the HF tokenizer call of the reference
        (hugging_face_model.py:179-185: padding=True, right-padded).  Only mask==1 tokens are run."""
        if ids.shape != attention_mask.shape or ids.ndim != 2:
            raise ValueError("ids and attention_mask must both be [n, S]")
        ids_h = _host_i64(ids)
        mask_h = _host_i64(attention_mask)
        n, S = ids_h.shape
        lengths = mask_h.sum(axis=1)
        if bool((lengths < 1).any()):
            raise ValueError("every sequence needs at least one unmasked token")
        prefix = np.arange(S)[None, :] < lengths[:, None]
        if not bool(((mask_h != 0) == prefix).all()):
            raise ValueError("attention_mask must be right-padded (a prefix of ones per row)")
        if n and int(lengths.max()) > self.arch.max_pos:
            raise ValueError(f"sequence longer than max_position_embeddings={self.arch.max_pos}")
        small = self._small_call(ids_h, lengths, normalize, clip=False)
        if small is not None:
            return small
        if n == 1 and self._graphs_ok():
            one = self._encode_one(torch.from_numpy(ids_h[0, :int(lengths[0])]), normalize, clip=False)
            if one is not None:
                return one
        out = torch.empty(n, self.out_width, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device), _large_call(self.device, int(lengths.sum())):
            for a, b in self._chunks(lengths):
                packed, cu = _pack(ids_h[a:b], lengths[a:b])
                d_ids = self._to_device(packed)
                d_cu = self._to_device(cu)
                rows, nseq = packed.numel(), b - a
                need = self.lib.mq_bert_workspace_bytes(C.byref(self.cfg), rows, nseq)
                self._call_text(False, d_ids, d_cu, cu, nseq, None, out[a:b], normalize, self._workspace(need))
        return out

    def queue_rows_ids(self, ids_h: np.ndarray, mask_h: np.ndarray, normalize: bool = True) -> Optional[np.ndarray]:
        """`queue_rows` for the tokeniser's right-padded [n, S] ids + attention mask (the engine's own tokenisers: the mask is a prefix of ones by
        construction; encode_ids checks that for foreign callers)"""
        if ids_h.ndim != 2 or ids_h.shape != mask_h.shape:
            return None
        lengths = mask_h.sum(axis=1)
        if ids_h.shape[0] and (int(lengths.min()) < 1 or int(lengths.max()) > self.arch.max_pos):
            return None        # (the regular path raises the error with its text)
        return self.queue_rows(ids_h, lengths, normalize, clip=False)

    def encode_device(self, d_ids: Tensor, lengths: Tensor, normalize: bool = True) -> Tensor:
        """ids already on the device (engine/gpu_tokenizers.py): int32 [n, S] rows [CLS] ... [SEP] pad..., lengths int64 [n] on the
        host.  Same as encode_ids with the right-padded mask those lengths imply; packing runs on the GPU."""
        return self._encode_device(d_ids, lengths, self.arch.max_pos, normalize, clip=False)


class HfClipTextTower(BertTower):
    """Text tower of open_clip's CustomTextCLIP checkpoints with an HF encoder (open_clip/xlm-roberta-base-ViT-B-32,
    xlm-roberta-large-ViT-H-14): `text.transformer.*` = the Hugging Face XLM-RoBERTa encoder (run by the BERT tower: post-LN, positions
    from 2), open_clip's MeanPooler over the non-pad tokens, then `text.proj` = Linear -> GELU -> Linear (no biases) inside
    mq_encode_bert (mq_bert_cfg.proj_hidden / out_dim); OPEN_CLIP.encode_text L2-normalises as for every CLIP."""

    def __init__(self, arch, sd: Dict[str, Tensor], device: str, precision: str = "bf16"):
        t = "text.transformer."
        enc = {k[len(t):]: v for k, v in sd.items() if k.startswith(t)}
        if not enc:
            raise ValueError("checkpoint has no text.transformer.* tensors (an HF text tower was expected)")
        super().__init__(arch.bert, enc, device, pooling="mean", precision=precision)
        self.clip_arch = arch
        W, Hd, D = arch.bert.width, arch.proj_hidden, arch.out_dim
        h = self._h
        # hidden units beyond Hd (zero-padded to the GEMMs' multiple of 64) are GELU(0) = 0 and meet zero columns of the second layer
        Hp = _ceil64(Hd)
        p1 = torch.zeros(Hp, W)
        p1[:Hd] = _need(sd, "text.proj.0.weight", (Hd, W)).detach().float()
        b1 = torch.zeros(Hp)
        if "text.proj.0.bias" in sd:
            b1[:Hd] = sd["text.proj.0.bias"].detach().float()
        p2 = torch.zeros(D, Hp)
        p2[:, :Hd] = _need(sd, "text.proj.2.weight", (D, Hd)).detach().float()
        if "text.proj.2.bias" in sd:
            raise ValueError("a biased second projection layer is not supported (open_clip builds text.proj without biases)")
        self.w.proj1_w, self.w.proj1_b, self.w.proj2_w = h.bf16(p1), h.f32(b1), h.bf16(p2)
        self.cfg.proj_hidden, self.cfg.out_dim = Hp, D
        if precision == "bf16":
            self.tune_residual_default()

    def encode_padded(self, ids: Tensor, normalize: bool = True) -> Tensor:
        """ids int [n, <= ctx] padded with pad_id, as open_clip's HFTokenizer hands them over; attention mask = ids != pad_id
        (hf_model.py HFTextEncoder.forward)"""
        ids = ids.detach().to("cpu", torch.int64)
        ids_h = _host_i64(ids)
        return self.encode_ids(ids_h, (ids_h != self.clip_arch.pad_id).astype(np.int64), normalize=normalize)


class MclipTextTower(BertTower):
    """Text encoder of the reference's `multilingual_clip` loader (clip_utils.py:521-565: pt_multilingual_clip.MultilingualCLIP, third-party
    and un-vendored): `transformer.*` = a Hugging Face encoder (XLM-RoBERTa large, or LaBSE = BERT), the attention-masked mean of its last
    hidden state, then `LinearTransformation` = ONE biased Linear to the paired CLIP image tower's embedding width — inside mq_encode_bert
    (mq_bert_weights.proj1_w / proj1_b with proj2_w NULL).  The loader normalises afterwards."""

    def __init__(self, bert_arch, out_dim: int, sd: Dict[str, Tensor], device: str, precision: str = "bf16"):
        t = "transformer."
        enc = {k[len(t):]: v for k, v in sd.items() if k.startswith(t)}
        if not enc:
            raise ValueError("checkpoint has no transformer.* tensors (a MultilingualCLIP text encoder was expected)")
        super().__init__(bert_arch, enc, device, pooling="mean", precision=precision)
        W = bert_arch.width
        self.w.proj1_w = self._h.bf16(_need(sd, "LinearTransformation.weight", (out_dim, W)))
        self.w.proj1_b = self._h.f32(_need(sd, "LinearTransformation.bias", (out_dim,)))
        self.cfg.proj_hidden, self.cfg.out_dim = 0, out_dim
        if precision == "bf16":
            self.tune_residual_default()
