"""Tensor-level wrappers over the C ABI (no autograd here; see functional.py).

PyTorch is plumbing only: it owns device memory and the HIP stream.  Every wrapper checks
device / dtype / contiguity on the host before a kernel sees a pointer — a faulting kernel
can reset the whole GPU host (shape checks are cheap, resets are not).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

A_KMAJOR, B_KMAJOR = 1, 2
EPI_BIAS, EPI_GELU, EPI_DGELU, EPI_RESIDUAL, EPI_ACCUM, EPI_A_ROWSUM = 1, 2, 4, 8, 16, 32
LAYOUT_NT = A_KMAJOR | B_KMAJOR     # y = x W^T
LAYOUT_NN = A_KMAJOR                # dx = dy W
LAYOUT_TN = 0                       # dW = dy^T x


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous float32 CUDA tensor, got {t.dtype} {t.device} "
                         f"contiguous={t.is_contiguous()}")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class _Workspace:
    """Grow-only scratch buffer per device (the library never allocates).  ONE buffer per device, whatever the stream:
    every launch sequence of this package runs on one stream at a time (a side stream only inside GraphedStep's
    warm-up, ordered by stream waits), and a buffer keyed by stream would be (re)allocated INSIDE a HIP-graph capture
    — from the graph's private pool, cached here beyond the graph's life and handed to the next capture on the same
    stream handle (seen as a 2e-4 drift of a replayed trainer after graph -> eager -> graph).  Outgrown buffers are
    kept alive (`retired`): a captured graph may still hold their address."""

    def __init__(self):
        self.buf = {}
        self.retired = []

    def get(self, nbytes: int, device) -> Optional[torch.Tensor]:
        if nbytes == 0:
            return None
        key = device.index
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("workspace growth inside a HIP-graph capture: run the step eagerly once before "
                                   "capturing (GraphedStep does)")
            if b is not None:
                self.retired.append(b)
            b = torch.empty(max(nbytes, 2 * (b.numel() if b is not None else 0), 1 << 20), dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


_ws = _Workspace()
_ws_lanes = {0: _ws}


class workspace_lane:
    """Launch sequences that run CONCURRENTLY on different streams must not share scratch memory (split-K slabs,
    LayerNorm / column-sum partials): `with ops.workspace_lane(1): ...` gives the enclosed launches a workspace of their
    own (grow-only like lane 0; sized by an eager warm-up before any HIP-graph capture)."""

    def __init__(self, lane: int):
        self.lane = lane

    def __enter__(self):
        global _ws
        self.prev = _ws
        _ws = _ws_lanes.setdefault(self.lane, _Workspace())
        return self

    def __exit__(self, *exc):
        global _ws
        _ws = self.prev
        return False


# ------------------------------------------------------------------------------------------- GEMM

def gemm(a: torch.Tensor, b: torch.Tensor, layout: int, *, out: Optional[torch.Tensor] = None,
         bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         aux: Optional[torch.Tensor] = None, epilogue: int = 0, alpha: float = 1.0, split_k: int = 0,
         a_rowsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = epilogue(alpha * op(a) @ op(b)); see include/dclip_hip.h for layout / epilogue bits.
    `a_rowsum` (float[M], [K][M]-major A only) additionally receives sum_k A[m,k]: the bias gradient of a wgrad."""
    lib = _lib.load()
    _f32(a, "a"), _f32(b, "b")
    if a.dim() != 2 or b.dim() != 2:
        raise ValueError("gemm: 2-D operands")
    if layout & A_KMAJOR:
        M, K = a.shape
    else:
        K, M = a.shape
    if layout & B_KMAJOR:
        N, Kb = b.shape
    else:
        Kb, N = b.shape
    if K != Kb:
        raise ValueError(f"gemm: contraction mismatch {tuple(a.shape)} x {tuple(b.shape)} layout={layout}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _f32(out, "out")
    if tuple(out.shape) != (M, N):
        raise ValueError(f"gemm: out shape {tuple(out.shape)} != {(M, N)}")
    if bias is not None:
        epilogue |= EPI_BIAS
        if _f32(bias, "bias").numel() != N:
            raise ValueError("gemm: bias size")
    if residual is not None:
        epilogue |= EPI_RESIDUAL
        if tuple(_f32(residual, "residual").shape) != (M, N):
            raise ValueError("gemm: residual shape")
    if aux is not None and tuple(_f32(aux, "aux").shape) != (M, N):
        raise ValueError("gemm: aux shape")
    if a_rowsum is not None:
        if aux is not None or (layout & A_KMAJOR) or _f32(a_rowsum, "a_rowsum").numel() != M:
            raise ValueError("gemm: a_rowsum needs a [K][M]-major A, no aux, and M elements")
        epilogue |= EPI_A_ROWSUM
        aux = a_rowsum
    nbytes = lib.dclip_gemm_f32_workspace(M, N, K, layout, split_k)
    ws = _ws.get(nbytes, a.device)
    _lib.check(lib.dclip_gemm_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), _ptr(bias), _ptr(residual),
                                  _ptr(aux), M, N, K, a.shape[1], b.shape[1], N, layout, epilogue,
                                  float(alpha), split_k, _ptr(ws), nbytes, _stream()), "gemm_f32")
    return out


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    lib = _lib.load()
    _f32(x, "x")
    M, N = x.shape
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
        accumulate = False
    _f32(out, "out")
    if out.numel() != N:
        raise ValueError("colsum: out size")
    nbytes = lib.dclip_colsum_f32_workspace(M, N)
    ws = _ws.get(nbytes, x.device)
    _lib.check(lib.dclip_colsum_f32(x.data_ptr(), out.data_ptr(), M, N, N, int(accumulate), _ptr(ws), nbytes,
                                    _stream()), "colsum")
    return out


# ------------------------------------------------------------------------------------------- LayerNorm

def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, save_stats: bool = True):
    lib = _lib.load()
    _f32(x, "x"), _f32(gamma, "gamma"), _f32(beta, "beta")
    D = x.shape[-1]
    rows = x.numel() // D
    if gamma.numel() != D or beta.numel() != D:
        raise ValueError("layernorm: gamma/beta size")
    y = torch.empty_like(x)
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device) if save_stats else None
    _lib.check(lib.dclip_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ptr(mean),
                                       _ptr(rstd), rows, D, float(eps), _stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, *, dresidual=None, dgamma=None, dbeta=None, accumulate=False,
                  need_param_grads=True, want_bf16: bool = False, dx_colsum: Optional[torch.Tensor] = None):
    """Returns (dx, dgamma, dbeta); dgamma/dbeta are written (or accumulated into) if requested.
    bf16 training path: `want_bf16` -> returns (dx, dgamma, dbeta, dx16) with the bf16 copy of dx from the same pass;
    `dx_colsum` [D] receives the column sums of dx (the bias gradient of the Linear that produced LayerNorm's input)."""
    lib = _lib.load()
    _f32(dy, "dy"), _f32(x, "x"), _f32(gamma, "gamma"), _f32(mean, "mean"), _f32(rstd, "rstd")
    D = x.shape[-1]
    rows = x.numel() // D
    if dy.shape != x.shape or mean.numel() != rows or rstd.numel() != rows or gamma.numel() != D:
        raise ValueError("layernorm_bwd: shape mismatch")
    if dresidual is not None and _f32(dresidual, "dresidual").shape != x.shape:
        raise ValueError("layernorm_bwd: dresidual shape")
    dx = torch.empty_like(x)
    if need_param_grads:
        if dgamma is None:
            dgamma = torch.empty_like(gamma)
            dbeta = torch.empty_like(gamma)
            accumulate = False
        _f32(dgamma, "dgamma"), _f32(dbeta, "dbeta")
        if dgamma.numel() != D or dbeta.numel() != D:
            raise ValueError("layernorm_bwd: dgamma/dbeta size")
    else:
        dgamma = dbeta = None
    if dx_colsum is not None and (_f32(dx_colsum, "dx_colsum").numel() != D):
        raise ValueError("layernorm_bwd: dx_colsum size")
    nbytes = lib.dclip_layernorm_bwd_workspace(rows, D) if (need_param_grads or dx_colsum is not None) else 0
    ws = _ws.get(nbytes, x.device)
    if not want_bf16 and dx_colsum is None:
        _lib.check(lib.dclip_layernorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                           rstd.data_ptr(), _ptr(dresidual), dx.data_ptr(), _ptr(dgamma), _ptr(dbeta),
                                           rows, D, int(accumulate), _ptr(ws), nbytes, _stream()), "layernorm_bwd")
        return dx, dgamma, dbeta
    dx16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    _lib.check(lib.dclip_layernorm_bwd_ex(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                          _ptr(dresidual), dx.data_ptr(), _ptr(dx16), _ptr(dgamma), _ptr(dbeta),
                                          _ptr(dx_colsum), rows, D, int(accumulate), _ptr(ws), nbytes, _stream()),
               "layernorm_bwd_ex")
    return (dx, dgamma, dbeta, dx16) if want_bf16 else (dx, dgamma, dbeta)


# ------------------------------------------------------------------------------------------- attention

def attention_fwd(qkv: torch.Tensor, B: int, S: int, H: int, causal: bool):
    lib = _lib.load()
    _f32(qkv, "qkv")
    if qkv.numel() != B * S * 3 * H * 64:
        raise ValueError(f"attention_fwd: qkv has {qkv.numel()} elements, expected B*S*3*H*64 = {B * S * 3 * H * 64}")
    out = torch.empty((B * S, H * 64), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.dclip_attention_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, S, H, int(causal),
                                       _stream()), "attention_fwd")
    return out, lse


def attention_bwd(qkv, out, dout, lse, B: int, S: int, H: int, causal: bool):
    lib = _lib.load()
    _f32(qkv, "qkv"), _f32(out, "out"), _f32(dout, "dout"), _f32(lse, "lse")
    if qkv.numel() != B * S * 3 * H * 64 or out.numel() != B * S * H * 64 or dout.numel() != out.numel() \
            or lse.numel() != B * H * S:
        raise ValueError("attention_bwd: shape mismatch")
    dqkv = torch.empty_like(qkv)
    # delta, and for long non-causal sequences the dS blocks the dK/dV kernel hands to the dQ kernel (include/dclip_hip.h)
    nbytes = int(lib.dclip_attention_bwd_workspace(B, S, H, int(causal)))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=qkv.device)
    _lib.check(lib.dclip_attention_bwd_ws(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                          dqkv.data_ptr(), ws.data_ptr(), nbytes, B, S, H, int(causal), _stream()),
               "attention_bwd")
    return dqkv


def attention_cls_fwd(qkv: torch.Tensor, B: int, S: int, H: int):
    """Attention output of query row 0 of every image only: ([B, H*64], lse [B, H])."""
    lib = _lib.load()
    _f32(qkv, "qkv")
    if qkv.numel() != B * S * 3 * H * 64:
        raise ValueError("attention_cls_fwd: qkv size")
    out = torch.empty((B, H * 64), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B, H), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.dclip_attention_cls_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, S, H, _stream()),
               "attention_cls_fwd")
    return out, lse


def attention_row_fwd(qkv: torch.Tensor, rows: torch.Tensor, B: int, S: int, H: int):
    """Causal attention output of ONE row per batch (sequence position rows[b]): [B, H*64]."""
    lib = _lib.load()
    _f32(qkv, "qkv"), _idx(rows, B)
    if qkv.numel() != B * S * 3 * H * 64:
        raise ValueError("attention_row_fwd: qkv size")
    out = torch.empty((B, H * 64), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B, H), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.dclip_attention_row_fwd(qkv.data_ptr(), rows.data_ptr(), out.data_ptr(), lse.data_ptr(), B, S, H,
                                           _stream()), "attention_row_fwd")
    return out


def attention_cls_bwd(qkv, out, dout, lse, B: int, S: int, H: int):
    lib = _lib.load()
    _f32(qkv, "qkv"), _f32(out, "out"), _f32(dout, "dout"), _f32(lse, "lse")
    if qkv.numel() != B * S * 3 * H * 64 or out.numel() != B * H * 64 or dout.numel() != out.numel() \
            or lse.numel() != B * H:
        raise ValueError("attention_cls_bwd: shape mismatch")
    dqkv = torch.empty_like(qkv)
    _lib.check(lib.dclip_fill(dqkv.data_ptr(), 0.0, dqkv.numel(), _stream()), "fill")
    delta = torch.empty((B * H,), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.dclip_attention_cls_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                           dqkv.data_ptr(), delta.data_ptr(), B, S, H, _stream()), "attention_cls_bwd")
    return dqkv


# ------------------------------------------------------------------------------------------- embeddings

def im2col(pixels: torch.Tensor, patch: int) -> torch.Tensor:
    lib = _lib.load()
    _f32(pixels, "pixel_values")
    B, Cc, Hh, Ww = pixels.shape
    if Hh != Ww or Hh % patch:
        raise ValueError(f"im2col: image {Hh}x{Ww} not a multiple of patch {patch}")
    g = Hh // patch
    cols = torch.empty((B * g * g, Cc * patch * patch), dtype=torch.float32, device=pixels.device)
    _lib.check(lib.dclip_im2col(pixels.data_ptr(), cols.data_ptr(), B, Cc, Hh, Ww, patch, _stream()), "im2col")
    return cols


def im2col_bf16(pixels: torch.Tensor, patch: int) -> torch.Tensor:
    """im2col with a bf16 result (rows padded to a multiple of 8 columns, zero filled); patch % 4 == 0."""
    lib = _lib.load()
    _f32(pixels, "pixel_values")
    B, Cc, Hh, Ww = pixels.shape
    if Hh != Ww or Hh % patch or patch % 4:
        raise ValueError(f"im2col_bf16: image {Hh}x{Ww}, patch {patch}")
    g, kdim = Hh // patch, Cc * patch * patch
    ld = (kdim + 7) // 8 * 8
    alloc = torch.zeros if ld != kdim else torch.empty
    cols = alloc((B * g * g, ld), dtype=torch.bfloat16, device=pixels.device)
    _lib.check(lib.dclip_im2col_bf16(pixels.data_ptr(), cols.data_ptr(), B, Cc, Hh, Ww, patch, ld, _stream()), "im2col_bf16")
    return cols


def vision_assemble_fwd(patch_emb, cls, pos, B: int, S: int, D: int) -> torch.Tensor:
    lib = _lib.load()
    _f32(patch_emb, "patch_emb"), _f32(cls, "class_embedding"), _f32(pos, "position_embedding")
    if patch_emb.numel() != B * (S - 1) * D or cls.numel() != D or pos.numel() != S * D:
        raise ValueError("vision_assemble_fwd: shape mismatch")
    x = torch.empty((B * S, D), dtype=torch.float32, device=patch_emb.device)
    _lib.check(lib.dclip_vision_assemble_fwd(patch_emb.data_ptr(), cls.data_ptr(), pos.data_ptr(), x.data_ptr(),
                                             B, S, D, _stream()), "vision_assemble_fwd")
    return x


def vision_assemble_bwd(dx, B: int, S: int, D: int) -> torch.Tensor:
    lib = _lib.load()
    _f32(dx, "dx")
    if dx.numel() != B * S * D:
        raise ValueError("vision_assemble_bwd: shape mismatch")
    dpatch = torch.empty((B * (S - 1), D), dtype=torch.float32, device=dx.device)
    _lib.check(lib.dclip_vision_assemble_bwd(dx.data_ptr(), dpatch.data_ptr(), B, S, D, _stream()),
               "vision_assemble_bwd")
    return dpatch


def _ids(ids: torch.Tensor) -> torch.Tensor:
    if not (ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous() and ids.dim() == 2):
        raise ValueError("input_ids: expected a contiguous int64 CUDA tensor [B, T]")
    return ids


def text_embed_fwd(ids, tok, pos) -> torch.Tensor:
    lib = _lib.load()
    _ids(ids), _f32(tok, "token_embedding"), _f32(pos, "position_embedding")
    B, T = ids.shape
    vocab, D = tok.shape
    if pos.shape[0] < T or pos.shape[1] != D:
        raise ValueError(f"text_embed_fwd: sequence {T} longer than position table {tuple(pos.shape)}")
    x = torch.empty((B * T, D), dtype=torch.float32, device=tok.device)
    _lib.check(lib.dclip_text_embed_fwd(ids.data_ptr(), tok.data_ptr(), pos.data_ptr(), x.data_ptr(), B, T, D, vocab,
                                        _stream()), "text_embed_fwd")
    return x


def text_embed_bwd(ids, dx, dtok):
    lib = _lib.load()
    _ids(ids), _f32(dx, "dx"), _f32(dtok, "dtok")
    B, T = ids.shape
    vocab, D = dtok.shape
    if dx.numel() != B * T * D:
        raise ValueError("text_embed_bwd: shape mismatch")
    _lib.check(lib.dclip_text_embed_bwd(ids.data_ptr(), dx.data_ptr(), dtok.data_ptr(), B, T, D, vocab, _stream()),
               "text_embed_bwd")
    return dtok


def first_eos(ids, eos_id: int) -> torch.Tensor:
    lib = _lib.load()
    _ids(ids)
    B, T = ids.shape
    idx = torch.empty((B,), dtype=torch.int32, device=ids.device)
    _lib.check(lib.dclip_first_eos(ids.data_ptr(), idx.data_ptr(), B, T, int(eos_id), _stream()), "first_eos")
    return idx


def _idx(idx, B):
    if idx is None:
        return None
    if not (idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous() and idx.numel() == B):
        raise ValueError("row index: expected contiguous int32 CUDA tensor [B]")
    return idx


def gather_rows(x, idx, B: int, S: int, D: int) -> torch.Tensor:
    lib = _lib.load()
    _f32(x, "x"), _idx(idx, B)
    if x.numel() != B * S * D:
        raise ValueError("gather_rows: shape mismatch")
    out = torch.empty((B, D), dtype=torch.float32, device=x.device)
    _lib.check(lib.dclip_gather_rows(x.data_ptr(), _ptr(idx), out.data_ptr(), B, S, D, _stream()), "gather_rows")
    return out


def scatter_rows(dout, idx, B: int, S: int, D: int) -> torch.Tensor:
    lib = _lib.load()
    _f32(dout, "dout"), _idx(idx, B)
    if dout.numel() != B * D:
        raise ValueError("scatter_rows: shape mismatch")
    dx = torch.empty((B * S, D), dtype=torch.float32, device=dout.device)
    _lib.check(lib.dclip_scatter_rows(dout.data_ptr(), _ptr(idx), dx.data_ptr(), B, S, D, _stream()), "scatter_rows")
    return dx


# ------------------------------------------------------------------------------------------- losses

NORM_EPS = 1e-12


def normalize_rows_fwd(x):
    lib = _lib.load()
    _f32(x, "x")
    B, Pd = x.shape
    xhat = torch.empty_like(x)
    inv = torch.empty((B,), dtype=torch.float32, device=x.device)
    _lib.check(lib.dclip_normalize_rows_fwd(x.data_ptr(), xhat.data_ptr(), inv.data_ptr(), B, Pd, NORM_EPS, _stream()),
               "normalize_rows_fwd")
    return xhat, inv


def normalize_rows_bwd(dxhat, xhat, inv, dx=None, accumulate=False):
    lib = _lib.load()
    _f32(dxhat, "dxhat"), _f32(xhat, "xhat"), _f32(inv, "inv")
    B, Pd = xhat.shape
    if dxhat.shape != xhat.shape or inv.numel() != B:
        raise ValueError("normalize_rows_bwd: shape mismatch")
    if dx is None:
        dx = torch.empty_like(xhat)
        accumulate = False
    _f32(dx, "dx")
    _lib.check(lib.dclip_normalize_rows_bwd(dxhat.data_ptr(), xhat.data_ptr(), inv.data_ptr(), dx.data_ptr(), B, Pd,
                                            NORM_EPS, int(accumulate), _stream()), "normalize_rows_bwd")
    return dx


def contrastive_lse(a_local, b_global, offset: int, inv_temp: float):
    lib = _lib.load()
    _f32(a_local, "a_local"), _f32(b_global, "b_global")
    Bl, Pd = a_local.shape
    Bg, Pb = b_global.shape
    if Pd != Pb or not (0 <= offset and offset + Bl <= Bg):
        raise ValueError(f"contrastive_lse: shapes {tuple(a_local.shape)} {tuple(b_global.shape)} offset {offset}")
    lse = torch.empty((Bl,), dtype=torch.float32, device=a_local.device)
    diag = torch.empty((Bl,), dtype=torch.float32, device=a_local.device)
    nbytes = lib.dclip_contrastive_workspace(Bl, Bg, Pd)
    ws = _ws.get(nbytes, a_local.device)
    _lib.check(lib.dclip_contrastive_lse(a_local.data_ptr(), b_global.data_ptr(), lse.data_ptr(), diag.data_ptr(), Bl,
                                         Bg, Pd, offset, float(inv_temp), _ptr(ws), nbytes, _stream()),
               "contrastive_lse")
    return lse, diag


def contrastive_grad(a_local, b_global, lse_row, lse_col, offset: int, inv_temp: float, coef: float):
    lib = _lib.load()
    _f32(a_local, "a_local"), _f32(b_global, "b_global"), _f32(lse_row, "lse_row"), _f32(lse_col, "lse_col")
    Bl, Pd = a_local.shape
    Bg = b_global.shape[0]
    if lse_row.numel() != Bl or lse_col.numel() != Bg or b_global.shape[1] != Pd:
        raise ValueError("contrastive_grad: shape mismatch")
    da = torch.empty_like(a_local)
    nbytes = lib.dclip_contrastive_workspace(Bl, Bg, Pd)
    ws = _ws.get(nbytes, a_local.device)
    _lib.check(lib.dclip_contrastive_grad(a_local.data_ptr(), b_global.data_ptr(), lse_row.data_ptr(),
                                          lse_col.data_ptr(), da.data_ptr(), Bl, Bg, Pd, offset, float(inv_temp),
                                          float(coef), _ptr(ws), nbytes, _stream()), "contrastive_grad")
    return da


def cosine_loss_fwd(s, t):
    lib = _lib.load()
    _f32(s, "student"), _f32(t, "teacher")
    if s.shape != t.shape:
        raise ValueError(f"cosine_distillation_loss: shapes {tuple(s.shape)} vs {tuple(t.shape)} — the reference "
                         "raises here too (CLIP_image_distillation.py:573)")
    B, Pd = s.shape
    loss_sum = torch.empty((), dtype=torch.float32, device=s.device)
    cos = torch.empty((B,), dtype=torch.float32, device=s.device)
    _lib.check(lib.dclip_cosine_loss_fwd(s.data_ptr(), t.data_ptr(), loss_sum.data_ptr(), cos.data_ptr(), B, Pd,
                                         _stream()), "cosine_loss_fwd")
    return loss_sum, cos


def cosine_loss_bwd(s, t, cos, coef: float, ds=None, accumulate=False):
    lib = _lib.load()
    _f32(s, "student"), _f32(t, "teacher"), _f32(cos, "cos")
    B, Pd = s.shape
    if ds is None:
        ds = torch.empty_like(s)
        accumulate = False
    _lib.check(lib.dclip_cosine_loss_bwd(s.data_ptr(), t.data_ptr(), cos.data_ptr(), ds.data_ptr(), B, Pd, float(coef),
                                         int(accumulate), _stream()), "cosine_loss_bwd")
    return ds


def sub_reduce(a, b, scale: float, out=None, accumulate=False):
    lib = _lib.load()
    _f32(a, "a")
    if b is not None and _f32(b, "b").numel() != a.numel():
        raise ValueError("sub_reduce: size mismatch")
    if out is None:
        out = torch.empty((), dtype=torch.float32, device=a.device)
        accumulate = False
    _lib.check(lib.dclip_sub_reduce(a.data_ptr(), _ptr(b), out.data_ptr(), a.numel(), float(scale), int(accumulate),
                                    _stream()), "sub_reduce")
    return out


# ------------------------------------------------------------------------------------------- meta-teacher tail

def cross_attention_fwd(q, kv, B: int, Lq: int, Lk: int, H: int):
    lib = _lib.load()
    _f32(q, "q"), _f32(kv, "kv")
    E = H * 64
    if q.numel() != B * Lq * E or kv.numel() != B * Lk * 2 * E:
        raise ValueError(f"cross_attention_fwd: q {tuple(q.shape)} kv {tuple(kv.shape)} vs B={B} Lq={Lq} Lk={Lk} H={H}")
    out = torch.empty((B * Lq, E), dtype=torch.float32, device=q.device)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device)
    _lib.check(lib.dclip_cross_attention_fwd(q.data_ptr(), kv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, Lq, Lk, H,
                                             _stream()), "cross_attention_fwd")
    return out, lse


def cross_attention_bwd(q, kv, out, dout, lse, B: int, Lq: int, Lk: int, H: int):
    lib = _lib.load()
    for n, t in (("q", q), ("kv", kv), ("out", out), ("dout", dout), ("lse", lse)):
        _f32(t, n)
    E = H * 64
    if q.numel() != B * Lq * E or kv.numel() != B * Lk * 2 * E or out.numel() != q.numel() \
            or dout.numel() != q.numel() or lse.numel() != B * H * Lq:
        raise ValueError("cross_attention_bwd: shape mismatch")
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    delta = torch.empty((B * H * Lq,), dtype=torch.float32, device=q.device)
    _lib.check(lib.dclip_cross_attention_bwd(q.data_ptr(), kv.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                             lse.data_ptr(), dq.data_ptr(), dkv.data_ptr(), delta.data_ptr(), B, Lq, Lk,
                                             H, _stream()), "cross_attention_bwd")
    return dq, dkv


def aggregation_fwd(x, temperature: float = 2.0, out=None, out_scale: float = 1.0, accumulate: bool = False):
    lib = _lib.load()
    _f32(x, "x")
    B, L, E = x.shape
    if out is None:
        out = torch.empty((B, E), dtype=torch.float32, device=x.device)
        accumulate = False
    _f32(out, "out")
    w = torch.empty((B, L), dtype=torch.float32, device=x.device)
    _lib.check(lib.dclip_aggregation_fwd(x.data_ptr(), out.data_ptr(), w.data_ptr(), B, L, E, float(temperature),
                                         float(out_scale), int(accumulate), _stream()), "aggregation_fwd")
    return out, w


def aggregation_bwd(x, w, dout, temperature: float = 2.0, out_scale: float = 1.0):
    lib = _lib.load()
    _f32(x, "x"), _f32(w, "weights"), _f32(dout, "dout")
    B, L, E = x.shape
    if tuple(w.shape) != (B, L) or tuple(dout.shape) != (B, E):
        raise ValueError("aggregation_bwd: shape mismatch")
    dx = torch.empty_like(x)
    _lib.check(lib.dclip_aggregation_bwd(x.data_ptr(), w.data_ptr(), dout.data_ptr(), dx.data_ptr(), B, L, E,
                                         float(temperature), float(out_scale), _stream()), "aggregation_bwd")
    return dx


def pack_tokens(tokens, sentence, eos, Tmax: int):
    lib = _lib.load()
    _f32(tokens, "tokens"), _f32(sentence, "sentence")
    B, T, Pd = tokens.shape
    _idx(eos, B)
    if tuple(sentence.shape) != (B, Pd) or not (0 < Tmax <= T):
        raise ValueError("pack_tokens: shape mismatch")
    out = torch.empty((B, Tmax, Pd), dtype=torch.float32, device=tokens.device)
    _lib.check(lib.dclip_pack_tokens(tokens.data_ptr(), sentence.data_ptr(), eos.data_ptr(), out.data_ptr(), B, T, Tmax,
                                     Pd, _stream()), "pack_tokens")
    return out


def mask_rows(x, count):
    lib = _lib.load()
    _f32(x, "x")
    B, R, E = x.shape
    _idx(count, B)
    _lib.check(lib.dclip_mask_rows(x.data_ptr(), count.data_ptr(), B, R, E, _stream()), "mask_rows")
    return x


def sanitize_groups(x: torch.Tensor, rows: int, flags: Optional[torch.Tensor] = None):
    """x viewed as [groups, rows, E]: groups holding a NaN / Inf become zeros IN PLACE.  flags=None: detect (returns
    the int32 flags); flags given: zero the flagged groups (backward of the guard)."""
    lib = _lib.load()
    _f32(x, "x")
    E = x.shape[-1]
    total_rows = x.numel() // E
    if total_rows % rows:
        raise ValueError("sanitize_groups: rows must divide the row count")
    groups = total_rows // rows
    mode = 0 if flags is None else 1
    if flags is None:
        flags = torch.empty((groups,), dtype=torch.int32, device=x.device)
    _lib.check(lib.dclip_sanitize_groups(x.data_ptr(), flags.data_ptr(), groups, rows, E, mode, _stream()),
               "sanitize_groups")
    return x, flags


# ------------------------------------------------------------------------------------------- bf16 frozen-tower path

def _bf16(t: torch.Tensor, name: str) -> torch.Tensor:
    if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous bfloat16 CUDA tensor")
    return t


def cast_bf16(x: torch.Tensor, pad_to: int = 8, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[rows, cols] fp32 -> bf16 with the row length rounded up to `pad_to` (zero filled).  `out`: refresh an existing
    copy in place (the persistent bf16 weights of a training tower: a captured HIP graph keeps reading the same buffer)."""
    lib = _lib.load()
    _f32(x, "x")
    rows, cols = x.shape
    ld = (cols + pad_to - 1) // pad_to * pad_to
    if out is not None:
        if tuple(_bf16(out, "out").shape) != (rows, ld):
            raise ValueError(f"cast_bf16: out shape {tuple(out.shape)} != {(rows, ld)}")
        y = out
    else:
        y = torch.empty((rows, ld), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.dclip_cast_f32_bf16(x.data_ptr(), y.data_ptr(), rows, cols, cols, ld, _stream()), "cast_f32_bf16")
    return y


def layernorm_fwd_bf16(x, gamma, beta, eps: float, save_stats: bool = False):
    """nn.LayerNorm with fp32 statistics and a bf16 result; with `save_stats` returns (y, mean, rstd) for the backward."""
    lib = _lib.load()
    _f32(x, "x"), _f32(gamma, "gamma"), _f32(beta, "beta")
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    if not save_stats:
        _lib.check(lib.dclip_layernorm_fwd_bf16(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), rows, D,
                                                float(eps), _stream()), "layernorm_fwd_bf16")
        return y
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
    _lib.check(lib.dclip_layernorm_fwd_bf16_stats(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                                  mean.data_ptr(), rstd.data_ptr(), rows, D, float(eps), _stream()),
               "layernorm_fwd_bf16_stats")
    return y, mean, rstd


def transpose_bf16(x: torch.Tensor, want_copy: bool = False, out: Optional[torch.Tensor] = None):
    """x [rows, cols] fp32 or bf16 -> x^T [cols, ld] bf16 with ld = rows rounded up to 8 (zero padded): the
    token-contiguous operand of a weight-gradient GEMM.  `want_copy`: also the untransposed bf16 copy [rows, cols]
    (cols % 8 == 0) from the same pass.  Returns xT or (xT, copy)."""
    lib = _lib.load()
    if not (x.is_cuda and x.is_contiguous() and x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16)):
        raise ValueError("transpose_bf16: contiguous 2-D float32 / bfloat16 CUDA tensor")
    rows, cols = x.shape
    if cols % 4:
        raise ValueError("transpose_bf16: cols must be a multiple of 4")
    ld = (rows + 7) // 8 * 8
    if out is not None:
        if tuple(_bf16(out, "out").shape) != (cols, ld):
            raise ValueError(f"transpose_bf16: out shape {tuple(out.shape)} != {(cols, ld)}")
        yT = out
    else:
        yT = torch.empty((cols, ld), dtype=torch.bfloat16, device=x.device)
    copy = None
    if want_copy:
        if cols % 8:
            raise ValueError("transpose_bf16: the bf16 copy feeds a GEMM as A: cols must be a multiple of 8")
        copy = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.dclip_transpose_to_bf16(x.data_ptr(), int(x.dtype == torch.bfloat16), yT.data_ptr(), _ptr(copy), rows,
                                           cols, cols, ld, cols, _stream()), "transpose_to_bf16")
    return (yT, copy) if want_copy else yT


def _out_f32(out: Optional[torch.Tensor], shape, device, name: str) -> torch.Tensor:
    """`out` (a caller-named destination, e.g. a gradient's slice of its all-reduce bucket) checked, or a fresh tensor."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or not out.is_contiguous() or not out.is_cuda:
        raise ValueError(f"{name}: out must be a contiguous float32 CUDA tensor of shape {tuple(shape)}")
    return out


def mt_weights_table(recs):
    """Device table for mt_weights_bf16 from [(w fp32 [rows, cols...], w16 bf16 [rows, ld] or None, w16T bf16 [cols, ldT] or
    None)]; returns (table tensor, ntensors, total tiles).  One synchronous upload: build it once, outside graph capture."""
    import struct
    lib = _lib.load()
    assert lib.dclip_mt_weights_record_bytes() == 48
    blob, t0 = [], 0
    dev = None
    for w, w16, w16T in recs:
        _f32(w, "w")
        dev = w.device
        rows = w.shape[0]
        cols = w.numel() // rows
        ld = w16.shape[1] if w16 is not None else cols
        ldT = w16T.shape[1] if w16T is not None else rows
        if cols % 4 or ld % 4 or ldT % 8:
            raise ValueError("mt_weights_table: cols / ld must be multiples of 4, ldT of 8")
        if w16 is not None and (tuple(_bf16(w16, "w16").shape) != (rows, ld)):
            raise ValueError("mt_weights_table: w16 shape")
        if w16T is not None and (tuple(_bf16(w16T, "w16T").shape) != (cols, ldT)):
            raise ValueError("mt_weights_table: w16T shape")
        tiles_c = (max(cols, ld) + 63) // 64
        tiles_r = (max(rows, ldT) + 63) // 64
        blob.append(struct.pack("<QQQiiiiii", w.data_ptr(), _ptr(w16) or 0, _ptr(w16T) or 0, rows, cols, ld, ldT, t0, tiles_c))
        t0 += tiles_c * tiles_r
    raw = b"".join(blob)
    table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    return table, len(recs), t0


def mt_weights_bf16(table: torch.Tensor, ntensors: int, total_tiles: int) -> None:
    lib = _lib.load()
    _lib.check(lib.dclip_mt_weights_bf16(table.data_ptr(), ntensors, total_tiles, _stream()), "mt_weights_bf16")


def rowsum_bf16(x: torch.Tensor, n: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row sums (fp32) of the first n columns of a bf16 matrix [R, ld]."""
    lib = _lib.load()
    _bf16(x, "x")
    R, ld = x.shape
    n = ld if n is None else n
    out = _out_f32(out, (R,), x.device, "rowsum_bf16")
    _lib.check(lib.dclip_rowsum_bf16(x.data_ptr(), out.data_ptr(), R, n, ld, _stream()), "rowsum_bf16")
    return out


def attention_fwd_bf16(qkv: torch.Tensor, B: int, S: int, H: int, causal: bool) -> torch.Tensor:
    """qkv [B*S, 3*H*64] bf16 -> context [B*S, H*64] bf16 (frozen towers, forward only)."""
    lib = _lib.load()
    _bf16(qkv, "qkv")
    if tuple(qkv.shape) != (B * S, 3 * H * 64):
        raise ValueError(f"attention_fwd_bf16: qkv shape {tuple(qkv.shape)} != {(B * S, 3 * H * 64)}")
    out = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device=qkv.device)
    _lib.check(lib.dclip_attention_fwd_bf16(qkv.data_ptr(), out.data_ptr(), B, S, H, int(causal), _stream()),
               "attention_fwd_bf16")
    return out


def attention_fwd_bf16_lse(qkv: torch.Tensor, B: int, S: int, H: int, causal: bool):
    """bf16 MFMA attention forward that also returns the log-sum-exp (training): qkv [B*S, 3*H*64] bf16 -> (context bf16, lse fp32)."""
    lib = _lib.load()
    _bf16(qkv, "qkv")
    if tuple(qkv.shape) != (B * S, 3 * H * 64):
        raise ValueError(f"attention_fwd_bf16_lse: qkv shape {tuple(qkv.shape)} != {(B * S, 3 * H * 64)}")
    out = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B * H, S), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.dclip_attention_fwd_bf16_lse(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, S, H, int(causal), _stream()),
               "attention_fwd_bf16_lse")
    return out, lse


def attention_bwd_bf16(qkv, out, dout, lse, B: int, S: int, H: int, causal: bool) -> torch.Tensor:
    """Backward of attention_fwd_bf16_lse on the bf16 MFMAs (S <= 64): returns dqkv [B*S, 3*H*64] bf16."""
    lib = _lib.load()
    _bf16(qkv, "qkv"), _bf16(out, "out"), _bf16(dout, "dout"), _f32(lse, "lse")
    D = H * 64
    if tuple(qkv.shape) != (B * S, 3 * D) or tuple(out.shape) != (B * S, D) or tuple(dout.shape) != (B * S, D) \
            or lse.numel() != B * H * S:
        raise ValueError("attention_bwd_bf16: shape mismatch")
    dqkv = torch.empty_like(qkv)
    _lib.check(lib.dclip_attention_bwd_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                            B, S, H, int(causal), _stream()), "attention_bwd_bf16")
    return dqkv


def attention_row_fwd_bf16(qkv: torch.Tensor, rows: Optional[torch.Tensor], B: int, S: int, H: int) -> torch.Tensor:
    """One attention output row per sequence from a bf16 qkv [B*S, 3*H*64]: row 0 against all keys (rows None: the CLS row of a
    vision tower's last layer) or row rows[b] against keys 0..rows[b] (int32 [B]: the first-EOS row of the causal text tower).
    Returns [B, H*64] bf16."""
    lib = _lib.load()
    _bf16(qkv, "qkv")
    if tuple(qkv.shape) != (B * S, 3 * H * 64):
        raise ValueError(f"attention_row_fwd_bf16: qkv shape {tuple(qkv.shape)} != {(B * S, 3 * H * 64)}")
    if rows is not None and not (rows.is_cuda and rows.dtype == torch.int32 and rows.numel() == B and rows.is_contiguous()):
        raise ValueError("attention_row_fwd_bf16: rows must be a contiguous int32 CUDA tensor [B]")
    out = torch.empty((B, H * 64), dtype=torch.bfloat16, device=qkv.device)
    _lib.check(lib.dclip_attention_row_fwd_bf16(qkv.data_ptr(), None if rows is None else rows.data_ptr(), out.data_ptr(),
                                                B, S, H, _stream()), "attention_row_fwd_bf16")
    return out


def attention_fwd_io16(qkv: torch.Tensor, B: int, S: int, H: int, causal: bool):
    """Short sequences (S <= 80), bf16 in / bf16 out, fp32 arithmetic: qkv [B*S, 3*H*64] bf16 -> (context bf16, lse fp32)."""
    lib = _lib.load()
    _bf16(qkv, "qkv")
    if tuple(qkv.shape) != (B * S, 3 * H * 64):
        raise ValueError(f"attention_fwd_io16: qkv shape {tuple(qkv.shape)} != {(B * S, 3 * H * 64)}")
    out = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B * H, S), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.dclip_attention_fwd_io16(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, S, H, int(causal), _stream()),
               "attention_fwd_io16")
    return out, lse


def attention_bwd_io16(qkv, out, dout, lse, B: int, S: int, H: int, causal: bool) -> torch.Tensor:
    """Backward of attention_fwd_io16 (S <= 64): everything bf16 except lse; returns dqkv [B*S, 3*H*64] bf16."""
    lib = _lib.load()
    _bf16(qkv, "qkv"), _bf16(out, "out"), _bf16(dout, "dout"), _f32(lse, "lse")
    D = H * 64
    if tuple(qkv.shape) != (B * S, 3 * D) or tuple(out.shape) != (B * S, D) or tuple(dout.shape) != (B * S, D) \
            or lse.numel() != B * H * S:
        raise ValueError("attention_bwd_io16: shape mismatch")
    dqkv = torch.empty_like(qkv)
    _lib.check(lib.dclip_attention_bwd_io16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                            B, S, H, int(causal), _stream()), "attention_bwd_io16")
    return dqkv


def gemm_bf16(a: torch.Tensor, w: torch.Tensor, *, n: Optional[int] = None, k: Optional[int] = None,
              bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, gelu: bool = False,
              out_bf16: bool = False, save_preact: bool = False, dgelu_of: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None):
    """y = epilogue(a @ w^T): a [M, lda>=K] and w [N, ldw>=K] bf16 (K-major), fp32 accumulation; y fp32 or bf16.
    Training path: `save_preact` (with gelu) also returns the bf16 pre-activation -> (y, h); `dgelu_of=h` multiplies the
    result by quick_gelu'(h)."""
    lib = _lib.load()
    _bf16(a, "a"), _bf16(w, "w")
    M, lda = a.shape
    N, ldw = w.shape
    K = k if k is not None else min(lda, ldw)
    if n is not None:
        N = n
    epi = 0
    if bias is not None:
        epi |= EPI_BIAS
        if _f32(bias, "bias").numel() != N:
            raise ValueError("gemm_bf16: bias size")
    if gelu:
        epi |= EPI_GELU
    if residual is not None:
        epi |= EPI_RESIDUAL
        if tuple(_f32(residual, "residual").shape) != (M, N):
            raise ValueError("gemm_bf16: residual shape")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=a.device)
    elif tuple(out.shape) != (M, N) or out.dtype != (torch.bfloat16 if out_bf16 else torch.float32) or not out.is_contiguous():
        raise ValueError("gemm_bf16: out shape / dtype")
    aux = None
    if save_preact:
        if not gelu:
            raise ValueError("gemm_bf16: save_preact goes with gelu")
        aux = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    if dgelu_of is not None:
        if gelu or tuple(_bf16(dgelu_of, "dgelu_of").shape) != (M, N):
            raise ValueError("gemm_bf16: dgelu_of must be the [M, N] bf16 pre-activation (and excludes gelu)")
        epi |= EPI_DGELU
        aux = dgelu_of
    if aux is None:
        _lib.check(lib.dclip_gemm_bf16(a.data_ptr(), w.data_ptr(), out.data_ptr(), _ptr(bias), _ptr(residual), M, N, K, lda,
                                       ldw, N, epi, int(out_bf16), _stream()), "gemm_bf16")
    else:
        _lib.check(lib.dclip_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), out.data_ptr(), _ptr(bias), _ptr(residual),
                                          aux.data_ptr(), M, N, K, lda, ldw, N, epi, int(out_bf16), _stream()),
                   "gemm_bf16_ex")
    return (out, aux) if save_preact else out


def gemm_bf16_wgrad(a: torch.Tensor, w: torch.Tensor, k: int, splits: Optional[int] = None,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW [M, N] fp32 = a[:, :k] @ w[:, :k]^T over bf16 token-contiguous operands (a = dY^T [M, ld], w = X^T [N, ld]):
    split-K when the output has few tiles (deterministic: partials summed in fixed order)."""
    lib = _lib.load()
    _bf16(a, "a"), _bf16(w, "w")
    M, lda = a.shape
    N, ldw = w.shape
    if splits is None:
        splits = lib.dclip_gemm_bf16_splitk_plan(M, N, k)
    out = _out_f32(out, (M, N), a.device, "gemm_bf16_wgrad")
    nbytes = lib.dclip_gemm_bf16_splitk_workspace(M, N, splits)
    ws = _ws.get(nbytes, a.device)
    _lib.check(lib.dclip_gemm_bf16_splitk(a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, k, lda, ldw, N, splits,
                                          _ptr(ws), nbytes, _stream()), "gemm_bf16_splitk")
    return out


def gemm_bf16_wgrad_tokmajor(dy: torch.Tensor, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """dW [out, in] fp32 = dy^T x from the token-major bf16 operands dy [tokens, out], x [tokens, in] as the backward has
    them (no transposes).  Returns None when the library's token-major form does not apply to the shape (the caller then
    transposes and uses gemm_bf16_wgrad)."""
    lib = _lib.load()
    _bf16(dy, "dy"), _bf16(x, "x")
    K, M = dy.shape
    K2, N = x.shape
    if K != K2:
        raise ValueError("gemm_bf16_wgrad_tokmajor: token counts differ")
    splits = lib.dclip_gemm_bf16_wgrad_tokmajor_plan(M, N, K)
    if splits == 0:
        return None
    out = _out_f32(out, (M, N), dy.device, "gemm_bf16_wgrad_tokmajor")
    nbytes = lib.dclip_gemm_bf16_splitk_workspace(M, N, splits)
    ws = _ws.get(nbytes, dy.device)
    _lib.check(lib.dclip_gemm_bf16_wgrad_tokmajor(dy.data_ptr(), x.data_ptr(), out.data_ptr(), M, N, K, M, N, N, splits,
                                                  _ptr(ws), nbytes, _stream()), "gemm_bf16_wgrad_tokmajor")
    return out


def colsum_bf16(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Column sums (over the rows = tokens) of a bf16 matrix, fp32 result: a bias gradient from a bf16 dY."""
    lib = _lib.load()
    _bf16(x, "x")
    M, N = x.shape
    out = _out_f32(out, (N,), x.device, "colsum_bf16")
    nbytes = lib.dclip_colsum_f32_workspace(M, N)
    ws = _ws.get(nbytes, x.device)
    _lib.check(lib.dclip_colsum_bf16(x.data_ptr(), out.data_ptr(), M, N, N, 0, _ptr(ws), nbytes, _stream()), "colsum_bf16")
    return out


# ------------------------------------------------------------------------------------------- crop front end

def crop_resize(images_u8: torch.Tensor, dims: torch.Tensor, boxes: torch.Tensor, size: int,
                max_crop_h: int, max_crop_w: int) -> torch.Tensor:
    """images_u8 [B,Hmax,Wmax,3] uint8, dims [B,2] int32 (h,w), boxes [NR,5] int32 (b,x1,y1,x2,y2) -> [NR,3,S,S] fp32.
    max_crop_h/w must bound the boxes' extents (the caller has the box list on the host anyway)."""
    lib = _lib.load()
    if not (images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.is_contiguous() and images_u8.dim() == 4
            and images_u8.shape[3] == 3):
        raise ValueError("crop_resize: images must be a contiguous uint8 CUDA tensor [B,H,W,3]")
    B, Hmax, Wmax, _ = images_u8.shape
    for t, n, shape in ((dims, "dims", (B, 2)), (boxes, "boxes", (boxes.shape[0], 5))):
        if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and tuple(t.shape) == shape):
            raise ValueError(f"crop_resize: {n} must be a contiguous int32 CUDA tensor {shape}")
    NR = boxes.shape[0]
    if NR == 0 or max_crop_h <= 0 or max_crop_w <= 0:
        raise ValueError("crop_resize: empty box list")
    out = torch.empty((NR, 3, size, size), dtype=torch.float32, device=images_u8.device)
    nbytes = lib.dclip_crop_resize_workspace(NR, size, max_crop_h, max_crop_w)
    ws = _ws.get(nbytes, images_u8.device)
    _lib.check(lib.dclip_crop_resize_u8(images_u8.data_ptr(), dims.data_ptr(), boxes.data_ptr(), out.data_ptr(), B, Hmax,
                                        Wmax, NR, size, max_crop_h, max_crop_w, _ptr(ws), nbytes, _stream()),
               "crop_resize_u8")
    return out


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(images_u8: torch.Tensor, dims: torch.Tensor, size: int = 224, mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """images_u8 [B,Hmax,Wmax,3] uint8 (image b in the top-left dims[b]=(h,w) corner) -> pixel_values [B,3,S,S] fp32:
    shortest-edge BICUBIC resize, centre crop, 1/255, (x-mean)/std — bit-exact with HF CLIPImageProcessor (PIL)."""
    import ctypes
    lib = _lib.load()
    if not (images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.is_contiguous() and images_u8.dim() == 4
            and images_u8.shape[3] == 3):
        raise ValueError("clip_preprocess: images must be a contiguous uint8 CUDA tensor [B,H,W,3]")
    B, Hmax, Wmax, _ = images_u8.shape
    if not (dims.is_cuda and dims.dtype == torch.int32 and dims.is_contiguous() and tuple(dims.shape) == (B, 2)):
        raise ValueError("clip_preprocess: dims must be a contiguous int32 CUDA tensor [B,2]")
    out = torch.empty((B, 3, size, size), dtype=torch.float32, device=images_u8.device)
    nbytes = lib.dclip_clip_preprocess_workspace(B, Hmax, Wmax, size)
    ws = _ws.get(nbytes, images_u8.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    _lib.check(lib.dclip_clip_preprocess_u8(images_u8.data_ptr(), dims.data_ptr(), out.data_ptr(), B, Hmax, Wmax, size,
                                            ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p), _ptr(ws),
                                            nbytes, _stream()), "clip_preprocess_u8")
    return out


# ------------------------------------------------------------------------------------------- evaluation

def rowdot_gather(a, b, idx=None):
    lib = _lib.load()
    _f32(a, "a"), _f32(b, "b")
    Bq, Pd = a.shape
    Bk = b.shape[0]
    if b.shape[1] != Pd:
        raise ValueError("rowdot_gather: width mismatch")
    if idx is not None:
        _idx(idx, Bq)
    elif Bk < Bq:
        raise ValueError("rowdot_gather: idx=None needs Bk >= Bq")
    out = torch.empty((Bq,), dtype=torch.float32, device=a.device)
    _lib.check(lib.dclip_rowdot_gather(a.data_ptr(), b.data_ptr(), _ptr(idx), out.data_ptr(), Bq, Bk, Pd, _stream()),
               "rowdot_gather")
    return out


def rank_count(queries, candidates, thresh, gt=None):
    """count[i] = #{j != gt[i] : <queries_i, candidates_j> > thresh[i]}  (int32; gt=None means j != i)."""
    lib = _lib.load()
    _f32(queries, "queries"), _f32(candidates, "candidates"), _f32(thresh, "thresh")
    Bq, Pd = queries.shape
    Bk = candidates.shape[0]
    if candidates.shape[1] != Pd or thresh.numel() != Bq:
        raise ValueError("rank_count: shape mismatch")
    if gt is not None:
        _idx(gt, Bq)
    count = torch.empty((Bq,), dtype=torch.int32, device=queries.device)
    nbytes = lib.dclip_rank_count_workspace(Bq, Bk)
    ws = _ws.get(nbytes, queries.device)
    _lib.check(lib.dclip_rank_count(queries.data_ptr(), candidates.data_ptr(), thresh.data_ptr(), _ptr(gt),
                                    count.data_ptr(), Bq, Bk, Pd, _ptr(ws), nbytes, _stream()), "rank_count")
    return count


def axpby(x, y, a: float, b: float):
    """y = a*x + b*y (in place on y)."""
    lib = _lib.load()
    _f32(x, "x"), _f32(y, "y")
    if x.numel() != y.numel():
        raise ValueError("axpby: size mismatch")
    _lib.check(lib.dclip_axpby(x.data_ptr(), y.data_ptr(), float(a), float(b), x.numel(), _stream()), "axpby")
    return y
