"""Torch-tensor front end of the C ABI: allocates outputs/workspaces with torch (device memory
and streams are torch's job here), passes raw device pointers and the current stream.

Every function requires CUDA(HIP) tensors; nothing here computes on the host.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from ._lib import AdamItem, CellTailArgs, DistillArgs, DistillGrads, FavorPlan, FfnFwdArgs, ReduceJob, View4, WgradItem, check, lib

BF16 = torch.bfloat16
F32 = torch.float32

EPI_BF16, EPI_F32_BIAS_RES, EPI_GELU_DUAL, EPI_DGELU, EPI_F32, EPI_BF16_BIAS = range(6)

NB_FEATURES, NB_FEATURES_PAD, DIM_HEAD = 266, 288, 64


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda, "scmomer_amd ops need device tensors (no CPU path)"
    return t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a device tensor; scmomer_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")


def view4(t: torch.Tensor) -> View4:
    """[B,H,N,64] bf16 view with arbitrary b/h/n strides (d contiguous)."""
    if t.dtype != BF16 or t.dim() != 4 or t.shape[-1] != DIM_HEAD or t.stride(3) != 1:
        raise ValueError(f"expected bf16 [B,H,N,64] with contiguous last dim, got {t.dtype} {tuple(t.shape)} "
                         f"strides {t.stride()}")
    return View4(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def empty_rows(t: int, n: int, dtype, device) -> torch.Tensor:
    """[t, n] tensor whose row pitch is a multiple of 128 bytes (a view of a wider buffer when n is not): the
    64-column output chunks of the skinny GEMM then cover whole 128-byte lines, which avoids partial-line writes."""
    per = 128 // torch.empty((), dtype=dtype).element_size()
    ld = (n + per - 1) // per * per
    buf = torch.empty((t, ld), dtype=dtype, device=device)
    return buf if ld == n else buf[:, :n]


# ----------------------------------------------------------------------------- parameters
def cast_weight(w: torch.Tensor, want_plain=True, want_t=True, out=None) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """f32 [r,c] -> (bf16 [r,c], bf16 [c,r]); `out` = an earlier result pair to overwrite in place"""
    _req(w, F32, "weight")
    r, c = w.shape
    if out is not None:
        wb, wt = out
    else:
        wb = torch.empty((r, c), dtype=BF16, device=w.device) if want_plain else None
        wt = torch.empty((c, r), dtype=BF16, device=w.device) if want_t else None
    check(lib.scm_cast_weight(_p(w), r, c, _p(wb), _p(wt), _stream()), "scm_cast_weight")
    return wb, wt


def prep_omega(omega: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(omega, F32, "projection_matrix")
    if tuple(omega.shape) != (NB_FEATURES, DIM_HEAD):
        raise ValueError(f"projection_matrix must be [266,64], got {tuple(omega.shape)}")
    if out is None:
        out = torch.empty((NB_FEATURES_PAD, DIM_HEAD), dtype=BF16, device=omega.device)
    check(lib.scm_prep_omega(_p(omega), _p(out), _stream()), "scm_prep_omega")
    return out


# ----------------------------------------------------------------------------- embedding
def embed_fwd(tok: torch.Tensor, emb: torch.Tensor, g2v: Optional[torch.Tensor],
              err: Optional[torch.Tensor] = None) -> torch.Tensor:
    """err: device int32 scalar the caller owns; the kernel sets it to 1 when a token id is outside [0, V) (such a
    token is looked up as token 0) and never clears it.  Without one a fresh flag is attached to the result."""
    _req(tok, torch.int64, "tokens")
    _req(emb, F32, "token_emb.weight")
    b, n = tok.shape
    v, d = emb.shape
    if g2v is not None:
        _req(g2v, torch.float64, "pos_emb.emb.weight")
        if g2v.shape[0] < n or g2v.shape[1] != d:
            raise ValueError("positional table too small")
    out = torch.empty((b, n, d), dtype=F32, device=tok.device)
    if err is None:
        err = torch.zeros(1, dtype=torch.int32, device=tok.device)
    else:
        _req(err, torch.int32, "err")
    check(lib.scm_embed_fwd(_p(tok), _p(emb), _p(g2v), b, n, d, v, _p(out), _p(err), _stream()), "scm_embed_fwd")
    out._scm_err = err  # read by the caller where it synchronises anyway (no sync here)
    return out


def embed_bwd(tok: torch.Tensor, dx: torch.Tensor, num_tokens: int, grad: torch.Tensor, accumulate: bool,
              defer: Optional[list] = None):
    """defer: a list that collects the final reduction as a job for reduce_batch instead of launching it here"""
    t = tok.numel()
    d = dx.shape[-1]
    ws = torch.empty(lib.scm_embed_bwd_ws_floats(t, d, num_tokens), dtype=F32, device=dx.device)
    check(lib.scm_embed_bwd(_p(tok), _p(dx), t, d, num_tokens, _p(ws), None if defer is not None else _p(grad),
                            int(accumulate), _stream()), "scm_embed_bwd")
    if defer is not None:
        width = num_tokens * d
        defer.append(colsum_job(ws, ws.numel() // width, width, [(grad.view(-1), 0, width)], accumulate))


# ----------------------------------------------------------------------------- layer norm
def ln_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    _req(x, F32, "x")
    d = x.shape[-1]
    t = x.numel() // d
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    mean = torch.empty(t, dtype=F32, device=x.device)
    rstd = torch.empty(t, dtype=F32, device=x.device)
    check(lib.scm_ln_fwd(_p(x), _p(gamma), _p(beta), t, d, _p(y), _p(mean), _p(rstd), _stream()), "scm_ln_fwd")
    return y, mean, rstd


def ln_bwd(dy, x, mean, rstd, gamma, dres, dgamma, dbeta, dres_colsum, accumulate: bool, with_bf16: bool = False,
           defer: Optional[list] = None):
    """-> dx f32, or (dx, dx rounded to bf16) when with_bf16.  defer: see embed_bwd."""
    _req(dy, F32, "dy")
    d = x.shape[-1]
    t = x.numel() // d
    dx = torch.empty_like(x)
    dxb = torch.empty(x.shape, dtype=BF16, device=x.device) if with_bf16 else None
    ws = torch.empty(lib.scm_ln_bwd_ws_floats(t, d), dtype=F32, device=x.device)
    now = defer is None
    check(lib.scm_ln_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), t, d, _p(dx), _p(dxb), _p(ws),
                         _p(dgamma) if now else None, _p(dbeta) if now else None, _p(dres_colsum) if now else None,
                         int(accumulate), _stream()), "scm_ln_bwd")
    if not now:
        defer.append(colsum_job(ws, ws.numel() // (3 * d), 3 * d,
                                [(dgamma, 0, d), (dbeta, d, 2 * d), (dres_colsum, 2 * d, 3 * d)], accumulate))
    return (dx, dxb) if with_bf16 else dx


def gemm_res_ln_supported(a: torch.Tensor, w: torch.Tensor) -> bool:
    """whether y = a @ w^T + bias + resid and h = LN(y) run as one launch (gemm_res_ln) for these operands"""
    return bool(a.dtype == BF16 and w.dtype == BF16 and a.stride(1) == 1 and w.stride(1) == 1 and
                lib.scm_gemm_res_ln_supported(a.shape[0], w.shape[0], a.shape[1], a.stride(0), w.stride(0)))


def gemm_res_ln(a, w, bias, resid, gamma, beta):
    """-> y f32 [M,D] = a @ w^T + bias + resid, and (h bf16, mean, rstd) = the LayerNorm of y with gamma, beta: the
    output product of a sublayer and the next sublayer's normalisation in one launch"""
    m, k = a.shape
    d = w.shape[0]
    assert resid is None or (resid.is_contiguous() and resid.numel() == m * d)
    y = torch.empty((m, d), dtype=F32, device=a.device)
    h = torch.empty((m, d), dtype=BF16, device=a.device)
    mean = torch.empty(m, dtype=F32, device=a.device)
    rstd = torch.empty(m, dtype=F32, device=a.device)
    check(lib.scm_gemm_res_ln(_p(a), a.stride(0), _p(w), w.stride(0), m, d, k, _p(bias), _p(resid), _p(y), _p(gamma),
                              _p(beta), _p(h), _p(mean), _p(rstd), _stream()), "scm_gemm_res_ln")
    return y, (h, mean, rstd)


def gemm_ln_bwd_supported(a: torch.Tensor, w: torch.Tensor) -> bool:
    """whether dx = dres + LN'(a @ w^T) runs as one launch (gemm_ln_bwd) for these operands"""
    return bool(a.dtype == BF16 and w.dtype == BF16 and a.stride(1) == 1 and w.stride(1) == 1 and
                lib.scm_gemm_ln_bwd_supported(a.shape[0], w.shape[0], a.shape[1], a.stride(0), w.stride(0)))


def gemm_ln_bwd(a, w, x, mean, rstd, gamma, dres, dgamma, dbeta, dres_colsum, accumulate: bool,
                with_bf16: bool = False, defer: Optional[list] = None):
    """dx = dres + LN'(a @ w^T): the data-gradient product and the LayerNorm backward of a sublayer in one launch
    (the f32 product never reaches memory).  Arguments and result as ln_bwd, with dy = a[M,K] @ w[D,K]^T."""
    m, k = a.shape
    d = w.shape[0]
    assert x.shape[-1] == d and x.numel() == m * d and x.is_contiguous() and (dres is None or dres.is_contiguous())
    dx = torch.empty_like(x)
    dxb = torch.empty(x.shape, dtype=BF16, device=x.device) if with_bf16 else None
    ws = torch.empty(lib.scm_gemm_ln_bwd_ws_floats(m, d), dtype=F32, device=x.device)
    now = defer is None
    check(lib.scm_gemm_ln_bwd(_p(a), a.stride(0), _p(w), w.stride(0), m, d, k, _p(x), _p(mean), _p(rstd), _p(gamma),
                              _p(dres), _p(dx), _p(dxb), _p(ws), _p(dgamma) if now else None,
                              _p(dbeta) if now else None, _p(dres_colsum) if now else None, int(accumulate), _stream()),
          "scm_gemm_ln_bwd")
    if not now:
        defer.append(colsum_job(ws, ws.numel() // (3 * d), 3 * d,
                                [(dgamma, 0, d), (dbeta, d, 2 * d), (dres_colsum, 2 * d, 3 * d)], accumulate))
    return (dx, dxb) if with_bf16 else dx


# ----------------------------------------------------------------------------- linear products
def gemm_nt(epi: int, a: torch.Tensor, w: torch.Tensor, *, out0: torch.Tensor, out1=None, bias=None, resid=None,
            pre=None):
    """out0[M,N] = a[M,K] @ w[N,K]^T with epilogue `epi`; a is bf16 or f32 (row-major, last dim contiguous)."""
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k and w.dtype == BF16 and a.stride(1) == 1 and w.stride(1) == 1
    check(lib.scm_gemm_nt(epi, int(a.dtype == F32), _p(a), a.stride(0), _p(w), w.stride(0), m, n, k,
                          _p(out0), out0.stride(0), _p(out1), out1.stride(0) if out1 is not None else 0,
                          _p(bias), _p(resid), resid.stride(0) if resid is not None else 0,
                          _p(pre), pre.stride(0) if pre is not None else 0, _stream()), "scm_gemm_nt")
    return out0


# ----------------------------------------------------------------------------- FeedForward sublayer in one launch
def chain_pos(n: int) -> torch.Tensor:
    """position of unit u (u < n) in the chain order in which scm_ffn_chain_fwd reads its second weight (a bit
    permutation inside every block of 64 units; positions of a last, partial block reach up to its end)"""
    u = torch.arange(n)
    v = u & 63
    return (u & ~63) | (((v >> 2) & 1) << 5) | (((v >> 5) & 1) << 4) | (((v >> 3) & 1) << 3) | (((v >> 4) & 1) << 2) | (v & 3)


def chain_copy(w: torch.Tensor) -> torch.Tensor:
    """[R, F] -> [R, F rounded up to 64] with column u at its chain position, zeros elsewhere"""
    r, f = w.shape
    out = torch.zeros((r, (f + 63) // 64 * 64), dtype=w.dtype, device=w.device)
    out[:, chain_pos(f).to(w.device)] = w
    return out


def ffn_chain_supported(m: int, d: int, f: int) -> bool:
    return bool(lib.scm_ffn_chain_supported(m, d, f))


def ffn_chain_fwd(x, gamma, beta, w1, b1, w2c, b2, *, save: bool, next_ln=None):
    """x f32 [M,200]; w1 bf16 [F,200]; w2c bf16 [200,F] in chain order.  -> (y f32, tape or None, next or None) with
    tape = (h, mean, rstd, pre, act) and next = (h_next, mean_next, rstd_next) for next_ln = (gamma, beta)."""
    _req(x, F32, "x")
    m, d = x.shape
    f = w1.shape[0]
    dev = x.device
    y = torch.empty((m, d), dtype=F32, device=dev)
    a = FfnFwdArgs()
    a.M, a.D, a.F = m, d, f
    a.x, a.ldx, a.gamma, a.beta = _p(x), x.stride(0), _p(gamma), _p(beta)
    a.w1, a.ldw1, a.b1, a.w2c, a.ldw2c, a.b2 = _p(w1), w1.stride(0), _p(b1), _p(w2c), w2c.stride(0), _p(b2)
    tape = None
    if save:
        h = torch.empty((m, d), dtype=BF16, device=dev)
        mean = torch.empty(m, dtype=F32, device=dev)
        rstd = torch.empty(m, dtype=F32, device=dev)
        pre = empty_rows(m, f, BF16, dev)
        act = empty_rows(m, f, BF16, dev)
        a.h, a.ldh, a.mean, a.rstd = _p(h), h.stride(0), _p(mean), _p(rstd)
        a.pre, a.ldpre, a.act, a.ldact = _p(pre), pre.stride(0), _p(act), act.stride(0)
        tape = (h, mean, rstd, pre, act)
    a.y, a.ldy = _p(y), y.stride(0)
    nxt = None
    if next_ln is not None:
        hn = torch.empty((m, d), dtype=BF16, device=dev)
        mn = torch.empty(m, dtype=F32, device=dev)
        rn = torch.empty(m, dtype=F32, device=dev)
        a.gamma_next, a.beta_next = _p(next_ln[0]), _p(next_ln[1])
        a.h_next, a.ldh_next, a.mean_next, a.rstd_next = _p(hn), hn.stride(0), _p(mn), _p(rn)
        nxt = (hn, mn, rn)
    check(lib.scm_ffn_chain_fwd(C.byref(a), _stream()), "scm_ffn_chain_fwd")
    return y, tape, nxt


def wgrad_slices(t: int, p: int, q_bf16: bool = True) -> int:
    """token slices so that P-tiles * slices workgroups are about one per CU (256); every slice costs a
    [P,208] f32 slab of extra traffic, so no more than that.  P tiles are 128 wide for bf16 operands (the
    asynchronous-copy kernel) and 160 wide when the Q operand is f32."""
    tiles = max(1, (t + 63) // 64)
    ptiles = (p + 127) // 128 if q_bf16 else (p + 159) // 160
    return max(1, min(tiles, 256 // ptiles))


def wgrad(pm: torch.Tensor, qm: torch.Tensor, grad: torch.Tensor, bias_grad, trans: bool, accumulate: bool,
          defer: Optional[list] = None):
    """grad (+)= pm^T qm  ([P,Q], or its transpose when trans); bias_grad (+)= column sums of pm.
    defer: a list that collects the slab sum as a job for reduce_batch instead of launching it here."""
    t, p = pm.shape
    q = qm.shape[1]
    s = wgrad_slices(t, p, qm.dtype == BF16)
    ws = torch.empty(lib.scm_wgrad_ws_floats(p, s), dtype=F32, device=pm.device)
    check(lib.scm_wgrad(_p(pm), pm.stride(0), _p(qm), qm.stride(0), int(qm.dtype == F32), t, p, q, s, _p(ws),
                        None if defer is not None else _p(grad), _p(bias_grad), int(trans), int(accumulate), _stream()),
          "scm_wgrad")
    if defer is not None:
        j = ReduceJob()
        j.kind, j.accumulate, j.src, j.S, j.P, j.Q, j.trans = 0, int(accumulate), _p(ws), s, p, q, int(trans)
        j.out[0], j.out[1] = _p(grad), _p(bias_grad)
        defer.append((j, ws))


def wgrad_multi_supported(*mats) -> bool:
    """the shared launch needs bf16 operands with contiguous last dimension (any number of tokens: the kernel
    addresses each token slice from the slice's own first row)"""
    return all(m.dtype == BF16 and m.stride(1) == 1 for m in mats)


def wgrad_multi(products, accumulate: bool, defer: Optional[list] = None):
    """up to four weight gradients over the same tokens in one launch; products = [(pm, qm, grad, bias_grad, trans)]
    as in wgrad.  The slab sums go to `defer` (reduce_batch), or run right away as one batched reduction."""
    assert 1 <= len(products) <= 4
    t = products[0][0].shape[0]
    ptiles = sum((it[0].shape[1] + 127) // 128 for it in products)
    s = max(1, min((t + 63) // 64, 256 // ptiles))
    items, jobs = (WgradItem * len(products))(), []
    for it, (pm, qm, grad, bias_grad, trans) in zip(items, products):
        assert pm.shape[0] == t and qm.shape[0] == t
        p, q = pm.shape[1], qm.shape[1]
        ws = torch.empty(lib.scm_wgrad_ws_floats(p, s), dtype=F32, device=pm.device)
        it.Pm, it.ldp, it.Qm, it.ldq, it.P, it.Q = _p(pm), pm.stride(0), _p(qm), qm.stride(0), p, q
        it.with_bias, it.ws = int(bias_grad is not None), _p(ws)
        j = ReduceJob()
        j.kind, j.accumulate, j.src, j.S, j.P, j.Q, j.trans = 0, int(accumulate), _p(ws), s, p, q, int(trans)
        j.out[0], j.out[1] = _p(grad), _p(bias_grad)
        jobs.append((j, ws))
    check(lib.scm_wgrad_multi(items, len(products), t, s, _stream()), "scm_wgrad_multi")
    if defer is not None:
        defer.extend(jobs)
    else:
        reduce_batch(jobs)


def colsum_job(ws: torch.Tensor, rows: int, width: int, segs, accumulate: bool):
    """reduce_batch job: out[i] (+)= sum over the `rows` partial rows of ws[:, lo + i] for every (out, lo, hi) of segs
    whose out is not None"""
    j = ReduceJob()
    j.kind, j.accumulate, j.src, j.S, j.P = 1, int(accumulate), _p(ws), rows, width
    for k, (out, lo, hi) in enumerate(segs):
        j.out[k] = _p(out)
        j.lo[k], j.hi[k] = lo, hi
    return (j, ws)


def reduce_batch(jobs: list):
    """one launch for the collected reductions; the workspaces travel with the jobs and are released afterwards"""
    if not jobs:
        return
    arr = (ReduceJob * len(jobs))(*[j for j, _ in jobs])
    check(lib.scm_reduce_batch(arr, len(jobs), _stream()), "scm_reduce_batch")
    jobs.clear()


# ----------------------------------------------------------------------------- FAVOR+ attention
@dataclass
class FavorSaved:
    plan: FavorPlan
    ctx: torch.Tensor     # [BH,288,64] bf16
    z: torch.Tensor       # [BH,288]
    sums: torch.Tensor    # [BH,80]
    rowmax: torch.Tensor  # [BH*N]
    den: torch.Tensor     # [BH*N]
    kmax: torch.Tensor    # partial maxima
    kmax_idx: torch.Tensor  # their positions


def make_plan(b: int, h: int, n: int, group_cells: Optional[int] = None) -> FavorPlan:
    """group_cells: cells per key-stabiliser group (None: the whole batch is one group, i.e. one reference call)"""
    plan = FavorPlan()
    check(lib.scm_favor_make_plan(b, h, n, C.byref(plan)), "scm_favor_make_plan")
    if group_cells is not None and group_cells != b:
        if not 1 <= group_cells <= b:
            raise ValueError(f"group_cells must be in 1..{b}, got {group_cells}")
        # the context pass splits the tokens of a (cell, head) exactly as a call over one group alone would: every
        # cell's context sums then add up in the same order, and its attention output carries the same bits
        alone = FavorPlan()
        check(lib.scm_favor_make_plan(group_cells, h, n, C.byref(alone)), "scm_favor_make_plan")
        plan.ctx_chunks, plan.kmax_chunks = alone.ctx_chunks, alone.kmax_chunks
        plan.group_cells = group_cells
    return plan


def favor_fwd(q, k, v, omega_b, out, out_centered, group_cells: Optional[int] = None) -> FavorSaved:
    """q,k,v,out,out_centered: bf16 [B,H,N,64] views (any b/h/n strides).  out_centered=None: no backward will
    follow; the centred output and the per-token statistics are not written.  group_cells: see make_plan."""
    b, h, n, _ = q.shape
    dev = q.device
    plan = make_plan(b, h, n, min(group_cells, b) if group_cells else None)
    bh = b * h
    tape = out_centered is not None
    ctx = torch.empty((bh, NB_FEATURES_PAD, DIM_HEAD), dtype=BF16, device=dev)
    z = torch.empty((bh, NB_FEATURES_PAD), dtype=F32, device=dev)
    sums = torch.empty((bh, 80), dtype=F32, device=dev)
    rowmax = torch.empty(bh * n, dtype=F32, device=dev) if tape else None
    den = torch.empty(bh * n, dtype=F32, device=dev) if tape else None
    kmax = torch.empty(plan.kmax_chunks * bh, dtype=F32, device=dev)
    kidx = torch.empty(plan.kmax_chunks * bh, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.scm_favor_ws_floats(C.byref(plan)), dtype=F32, device=dev)
    check(lib.scm_favor_fwd(C.byref(plan), view4(q), view4(k), view4(v), _p(omega_b), view4(out),
                            view4(out_centered) if tape else View4(None, 0, 0, 0), _p(ctx), _p(z),
                            _p(sums), _p(rowmax), _p(den), _p(kmax), _p(kidx), _p(ws), _stream()), "scm_favor_fwd")
    return FavorSaved(plan, ctx, z, sums, rowmax, den, kmax, kidx)


def favor_bwd(q, k, v, o_centered, dout, omega_b, sv: FavorSaved, dq, dk, dv):
    dev = q.device
    plan = sv.plan
    bh = plan.B * plan.H
    dctx = torch.empty_like(sv.ctx)
    dz = torch.empty_like(sv.z)
    dsums = torch.empty_like(sv.sums)
    dden = torch.empty_like(sv.den)
    ws = torch.empty(lib.scm_favor_ws_floats(C.byref(plan)), dtype=F32, device=dev)
    check(lib.scm_favor_bwd(C.byref(plan), view4(q), view4(k), view4(v), view4(o_centered), view4(dout), _p(omega_b),
                            _p(sv.ctx), _p(sv.z), _p(sv.sums), _p(sv.rowmax), _p(sv.den), _p(sv.kmax),
                            _p(sv.kmax_idx), view4(dq), view4(dk), view4(dv), _p(dctx), _p(dz), _p(dsums), _p(dden), _p(ws),
                            _stream()), "scm_favor_bwd")


# ----------------------------------------------------------------------------- free functions (materialised, f32)
def feature_map(data: torch.Tensor, projection: torch.Tensor, is_query: bool, normalize_data: bool = True,
                eps: float = 1e-4) -> torch.Tensor:
    """softmax_kernel on the device: data f32 [..., N, 64] contiguous, projection f32 [M, 64] -> f32 [..., N, M]."""
    _req(data, F32, "data")
    _req(projection, F32, "projection_matrix")
    if data.shape[-1] != DIM_HEAD or projection.dim() != 2 or projection.shape[1] != DIM_HEAD:
        raise NotImplementedError(f"the MI355X kernels are built for a head dimension of 64 (got data "
                                  f"{tuple(data.shape)}, projection {tuple(projection.shape)})")
    n, m = data.shape[-2], projection.shape[0]
    bh = data.numel() // (n * DIM_HEAD) if data.numel() else 0
    out = torch.empty((*data.shape[:-1], m), dtype=F32, device=data.device)
    if out.numel() == 0:
        return out
    ws = None if is_query else torch.empty(lib.scm_feature_map_ws_floats(bh, n), dtype=F32, device=data.device)
    check(lib.scm_feature_map(_p(data), _p(projection), bh, n, m, int(bool(is_query)), int(bool(normalize_data)),
                              float(eps), _p(out), _p(ws), _stream()), "scm_feature_map")
    return out


def linear_attention(qf: torch.Tensor, kf: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """linear_attention on the device: q', k' f32 [..., N, M], v f32 [..., N, 64] -> f32 [..., N, 64]."""
    for t, nm in ((qf, "q"), (kf, "k"), (v, "v")):
        _req(t, F32, nm)
    if v.shape[-1] != DIM_HEAD or qf.shape != kf.shape or qf.shape[:-1] != v.shape[:-1]:
        raise NotImplementedError(f"expected q, k [..., N, M] and v [..., N, 64] (got {tuple(qf.shape)}, "
                                  f"{tuple(kf.shape)}, {tuple(v.shape)})")
    n, m = qf.shape[-2], qf.shape[-1]
    bh = v.numel() // (n * DIM_HEAD) if v.numel() else 0
    out = torch.empty(v.shape, dtype=F32, device=v.device)
    if out.numel() == 0:
        return out
    ws = torch.empty(lib.scm_linear_attention_ws_floats(bh, n, m), dtype=F32, device=v.device)
    check(lib.scm_linear_attention(_p(qf), _p(kf), _p(v), bh, n, m, _p(out), _p(ws), _stream()),
          "scm_linear_attention")
    return out


# ----------------------------------------------------------------------------- final norm + head
def head_fwd(x, gamma, beta, w, bias, relu: bool, want_enc: bool, want_out: bool):
    _req(x, F32, "x")
    d = x.shape[-1]
    t = x.numel() // d
    c = w.shape[0] if (w is not None and want_out) else 0
    enc = torch.empty_like(x) if want_enc else None
    out = torch.empty((*x.shape[:-1], c), dtype=F32, device=x.device) if c > 0 else None
    mean = torch.empty(t, dtype=F32, device=x.device)
    rstd = torch.empty(t, dtype=F32, device=x.device)
    check(lib.scm_head_fwd(_p(x), _p(gamma), _p(beta), _p(w) if c else None, _p(bias) if c else None, t, d, c,
                           int(relu), _p(enc), _p(out), _p(mean), _p(rstd), _stream()), "scm_head_fwd")
    return enc, out, mean, rstd


def head_bwd(dout, denc, out, x, mean, rstd, gamma, beta, w, relu: bool, dgamma, dbeta, dw, db, accumulate: bool,
             with_bf16: bool = False, defer: Optional[list] = None):
    """-> dx f32, or (dx, dx rounded to bf16) when with_bf16.  defer: see embed_bwd."""
    d = x.shape[-1]
    t = x.numel() // d
    c = w.shape[0] if (w is not None and dout is not None) else 0
    dx = torch.empty_like(x)
    dxb = torch.empty(x.shape, dtype=BF16, device=x.device) if with_bf16 else None
    ws = torch.empty(lib.scm_head_bwd_ws_floats(t, d, c), dtype=F32, device=x.device)
    now = defer is None
    check(lib.scm_head_bwd(_p(dout), _p(denc), _p(out), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta),
                           _p(w) if c else None, t, d, c, int(relu), _p(dx), _p(dxb), _p(ws),
                           _p(dgamma) if now else None, _p(dbeta) if now else None,
                           _p(dw) if (c and now) else None, _p(db) if (c and now) else None, int(accumulate), _stream()),
          "scm_head_bwd")
    if not now:
        width = 2 * d + c * d + c
        defer.append(colsum_job(ws, ws.numel() // width, width,
                                [(dgamma, 0, d), (dbeta, d, 2 * d), (dw.view(-1) if c else None, 2 * d, 2 * d + c * d),
                                 (db if c else None, 2 * d + c * d, width)], accumulate))
    return (dx, dxb) if with_bf16 else dx


# ----------------------------------------------------------------------------- S1 step tail
def s1_mask(tokens: torch.Tensor, rand_pick: torch.Tensor, rand_replace: torch.Tensor, mask_prob: float,
            replace_prob: float, pad_id: int, mask_id: int):
    """-> (masked_input, labels), int64 [B,N]"""
    import math
    _req(tokens, torch.int64, "tokens")
    _req(rand_pick, F32, "rand_pick")
    _req(rand_replace, F32, "rand_replace")
    b, n = tokens.shape
    masked = torch.empty_like(tokens)
    labels = torch.empty_like(tokens)
    check(lib.scm_s1_mask(_p(tokens), _p(rand_pick), _p(rand_replace), b, n, mask_prob, replace_prob,
                          math.ceil(mask_prob * n), pad_id, mask_id, _p(masked), _p(labels), _stream()), "scm_s1_mask")
    return masked, labels


def ce_fwd(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int, group_tokens: int = 0) -> torch.Tensor:
    """-> f32 [G, 2] = per loss group (mean loss over its counted tokens, number of counted tokens); the groups are
    runs of group_tokens consecutive tokens (0: one group over everything)"""
    _req(logits, F32, "logits")
    _req(labels, torch.int64, "labels")
    c = logits.shape[-1]
    t = logits.numel() // c
    ws = torch.empty(lib.scm_ce_ws_floats(t, group_tokens), dtype=F32, device=logits.device)
    out = torch.empty((lib.scm_ce_groups(t, group_tokens), 2), dtype=F32, device=logits.device)
    check(lib.scm_ce_fwd(_p(logits), _p(labels), t, c, ignore_index, group_tokens, _p(ws), _p(out), _stream()),
          "scm_ce_fwd")
    return out


def ce_bwd(logits, labels, ignore_index: int, dloss: torch.Tensor, loss_count: torch.Tensor,
           group_tokens: int = 0) -> torch.Tensor:
    """dloss: f32 [G], the incoming gradient of each group's loss"""
    c = logits.shape[-1]
    t = logits.numel() // c
    assert dloss.numel() == loss_count.shape[0] == lib.scm_ce_groups(t, group_tokens)
    dlogits = torch.empty_like(logits)
    check(lib.scm_ce_bwd(_p(logits), _p(labels), t, c, ignore_index, group_tokens, _p(dloss), _p(loss_count),
                         _p(dlogits), _stream()), "scm_ce_bwd")
    return dlogits


def s1_accuracy(logits: torch.Tensor, labels: torch.Tensor, pad_id: int):
    """-> (ratio f32 [B] = correct / counted per cell, counts int32 [B,2] = (correct, counted))"""
    _req(logits, F32, "logits")
    _req(labels, torch.int64, "labels")
    b, n, c = logits.shape
    ratio = torch.empty(b, dtype=F32, device=logits.device)
    buf = torch.empty(2 * b + lib.scm_s1_accuracy_ws_ints(b, n), dtype=torch.int32, device=logits.device)
    counts, ws = buf[:2 * b].view(b, 2), buf[2 * b:]
    check(lib.scm_s1_accuracy(_p(logits), _p(labels), b, n, c, pad_id, _p(ratio), _p(counts), _p(ws), _stream()),
          "scm_s1_accuracy")
    return ratio, counts


# ----------------------------------------------------------------------------- cell-embedding tail
def cell_tail_supported(p: int, hd: int, o: int) -> bool:
    return p <= 1024 and hd <= 1024 and o <= 1024 and p + hd <= 1024 and (p + hd) % 64 == 0


def cell_tail(genes, x, wp, ln_p, w1, b1, ln_h, w2, b2, wo, bo, normalize: bool) -> torch.Tensor:
    """scMomer.get_cell_embedding after the encoder (forward only, f32): genes, x [B,G]; wp [P,G]; ln_p = (w, b) of
    rna_ln; w1 [Hd,G], b1; ln_h = (w, b); w2 [P,Hd], b2; wo [O,2P], bo  ->  [B,O]"""
    tensors = (genes, x, wp, *ln_p, w1, b1, *ln_h, w2, b2, wo, bo)
    for t in tensors:
        _req(t, F32, "cell_tail operand")
    b, g = genes.shape
    p, hd, o = wp.shape[0], w1.shape[0], wo.shape[0]
    if x.shape != genes.shape or wp.shape[1] != g or w1.shape[1] != g or tuple(w2.shape) != (p, hd) or wo.shape[1] != 2 * p:
        raise ValueError("cell_tail: inconsistent shapes")
    a = CellTailArgs()
    a.P, a.Hd, a.O, a.normalize = p, hd, o, int(bool(normalize))
    a.Wp, a.ln_p_w, a.ln_p_b, a.W1, a.b1 = _p(wp), _p(ln_p[0]), _p(ln_p[1]), _p(w1), _p(b1)
    a.ln_h_w, a.ln_h_b, a.W2, a.b2, a.Wo, a.bo = _p(ln_h[0]), _p(ln_h[1]), _p(w2), _p(b2), _p(wo), _p(bo)
    ws = torch.empty(lib.scm_cell_tail_ws_floats(b, g, p, hd), dtype=F32, device=genes.device)
    out = torch.empty((b, o), dtype=F32, device=genes.device)
    check(lib.scm_cell_tail_fwd(_p(genes), _p(x), b, g, C.byref(a), _p(ws), _p(out), _stream()), "scm_cell_tail_fwd")
    return out


# ----------------------------------------------------------------------------- distillation tail
DISTILL_GRAD_ORDER = ("W1", "b1", "ln_w", "ln_b", "W2", "b2", "Wo", "bo", "Wa", "ba", "Wr", "br")


def distill_max_cells() -> int:
    return int(lib.scm_distill_max_cells())


def distill_supported(b: int, p: int, hd: int, o: int) -> bool:
    return cell_tail_supported(p, hd, o) and o == 128 and 0 < b <= lib.scm_distill_max_cells()


class DistillCall:
    """One forward of the S3 tail (scm_distill_fwd) and what its backward needs: the argument block, the tensors it
    points into, the tape and the scratch."""

    def __init__(self, genes, x, atac_raw, latent, drop, pos_weight, wp, ln_p, w1, b1, ln_h, w2, b2, wo, bo,
                 wa, ba, wr, br, normalize: bool, distill_weight: float = 1.0, group_cells: int = 0):
        tensors = [genes, x, atac_raw, latent, pos_weight, wp, *ln_p, w1, b1, *ln_h, w2, b2, wo, bo, wa, ba, wr, br]
        if drop is not None:
            tensors.append(drop)
        for t in tensors:
            _req(t, F32, "distill operand")
        b, g = genes.shape
        p, hd, o, na = wp.shape[0], w1.shape[0], wo.shape[0], wa.shape[0]
        if x.shape != genes.shape or wp.shape[1] != g or w1.shape[1] != g or tuple(w2.shape) != (p, hd) or \
                wo.shape[1] != 2 * p or tuple(wa.shape) != (na, o) or tuple(wr.shape) != (g, o) or \
                tuple(atac_raw.shape) != (b, na) or tuple(latent.shape) != (b, p) or ba.numel() != na or \
                br.numel() != g or pos_weight.numel() not in (1, na) or (drop is not None and tuple(drop.shape) != (b, hd)):
            raise ValueError("distill tail: inconsistent shapes")
        if not distill_supported(b, p, hd, o):
            raise ValueError(f"distill tail: unsupported sizes (cells {b}, widths {p}/{hd}/{o})")
        a = DistillArgs()
        t = a.tail
        t.P, t.Hd, t.O, t.normalize = p, hd, o, int(bool(normalize))
        t.Wp, t.ln_p_w, t.ln_p_b, t.W1, t.b1 = _p(wp), _p(ln_p[0]), _p(ln_p[1]), _p(w1), _p(b1)
        t.ln_h_w, t.ln_h_b, t.W2, t.b2, t.Wo, t.bo = _p(ln_h[0]), _p(ln_h[1]), _p(w2), _p(b2), _p(wo), _p(bo)
        a.B, a.G, a.Na, a.pos_weight_n, a.distill_weight = b, g, na, pos_weight.numel(), float(distill_weight)
        if group_cells and b % group_cells:
            raise ValueError(f"distill tail: {b} cells do not split into groups of {group_cells}")
        a.group_cells = int(group_cells or 0)
        self.groups = b // group_cells if group_cells else 1
        a.Wa, a.ba, a.Wr, a.br = _p(wa), _p(ba), _p(wr), _p(br)
        a.genes, a.x, a.atac_raw, a.latent, a.drop, a.pos_weight = _p(genes), _p(x), _p(atac_raw), _p(latent), _p(drop), \
            _p(pos_weight)
        self.args, self.keep = a, tensors
        self.shapes = {"W1": w1.shape, "b1": b1.shape, "ln_w": ln_h[0].shape, "ln_b": ln_h[1].shape, "W2": w2.shape,
                       "b2": b2.shape, "Wo": wo.shape, "bo": bo.shape, "Wa": wa.shape, "ba": ba.shape, "Wr": wr.shape,
                       "br": br.shape}
        dev = genes.device
        self.tape = torch.empty(lib.scm_distill_tape_floats(b, p, hd, o), dtype=F32, device=dev)
        self.ws = torch.empty(lib.scm_distill_ws_floats(b, g, p, hd, o, na), dtype=F32, device=dev)
        self.student = self.tape[2 * b * hd:2 * b * hd + b * 2 * p].view(b, 2 * p)[:, :p]   # the student's embedding

    def forward(self) -> torch.Tensor:
        """-> losses [groups, 4] = per loss group (BCE, RNA MSE, distillation MSE, BCE + RNA MSE + distill_weight *
        distillation MSE)"""
        losses = torch.empty((self.groups, 4), dtype=F32, device=self.tape.device)
        check(lib.scm_distill_fwd(C.byref(self.args), _p(self.tape), _p(self.ws), _p(losses), _stream()), "scm_distill_fwd")
        return losses

    def backward(self, gloss: torch.Tensor, grads=None, accumulate: bool = False):
        """Gradients of gloss . losses for the twelve trainable tensors in DISTILL_GRAD_ORDER: written into fresh
        tensors (grads None) or written / added into the given ones."""
        _req(gloss, F32, "distill gloss")
        if gloss.numel() != 3:
            raise ValueError("distill tail: gloss holds three values")
        if grads is None:
            grads = [torch.empty(tuple(self.shapes[n]), dtype=F32, device=gloss.device) for n in DISTILL_GRAD_ORDER]
            accumulate = False
        g = DistillGrads()
        for n, t in zip(DISTILL_GRAD_ORDER, grads):
            _req(t, F32, "distill gradient")
            if t.numel() != int(torch.Size(self.shapes[n]).numel()):
                raise ValueError(f"distill tail: gradient buffer of {n} has {t.numel()} elements")
            setattr(g, "d" + n, _p(t))
        check(lib.scm_distill_bwd(C.byref(self.args), _p(self.tape), _p(self.ws), _p(gloss), C.byref(g),
                                  int(bool(accumulate)), _stream()), "scm_distill_bwd")
        return grads


# ----------------------------------------------------------------------------- optimizer step
class AdamTable:
    """Device table of scm_adam_item chunks over (parameter, gradient, exp_avg, exp_avg_sq) quadruples: built once
    per set of tensors, reused by every adam_step (the tensors' storage must stay where it is)."""

    def __init__(self, params, grads, exp_avgs, exp_avg_sqs):
        import ctypes
        chunk = lib.scm_adam_chunk_elems()
        items = []
        self.keys = []
        for ti, (p, g, m, v) in enumerate(zip(params, grads, exp_avgs, exp_avg_sqs)):
            for t, nm in ((p, "parameter"), (g, "gradient"), (m, "exp_avg"), (v, "exp_avg_sq")):
                _req(t, F32, nm)
                if t.numel() != p.numel():
                    raise ValueError(f"{nm}: {t.numel()} elements for a parameter of {p.numel()}")
            self.keys.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()))
            for off in range(0, p.numel(), chunk):
                it = AdamItem()
                it.p, it.g, it.m, it.v = (t.data_ptr() + 4 * off for t in (p, g, m, v))
                it.n, it.tensor = min(chunk, p.numel() - off), ti
                items.append(it)
        arr = (AdamItem * len(items))(*items)
        raw = bytes(memoryview(arr).cast("B"))
        dev = params[0].device
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.n_chunks = len(items)
        self.ws = torch.empty(len(items) + 2, dtype=F32, device=dev)
        self.norm = torch.zeros(1, dtype=F32, device=dev)

    def matches(self, params, grads, exp_avgs, exp_avg_sqs) -> bool:
        return self.keys == [(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
                             for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs)]


def adam_step(table: AdamTable, *, step: int, lr: float, beta1: float, beta2: float, eps: float, max_norm: float,
              grad_scale: float = 1.0) -> torch.Tensor:
    """clip_grad_norm_(max_norm) + Adam.step() + zero_grad() over every tensor of the table in three launches; -> the
    gradient norm before clipping (device scalar, overwritten by the next call)"""
    check(lib.scm_adam_step(table.table.data_ptr(), table.n_chunks, _p(table.ws), grad_scale, max_norm, lr, beta1, beta2,
                            eps, step, _p(table.norm), _stream()), "scm_adam_step")
    return table.norm


# ----------------------------------------------------------------------------- ingest, projection redraw
RULE_S1, RULE_SCMOMER = 0, 1


def bin_csr(indptr: torch.Tensor, indices: torch.Tensor, values: torch.Tensor, rows: Optional[torch.Tensor],
            n_genes: int, rule: int, limit: int, err_flag: torch.Tensor, batch: Optional[int] = None) -> torch.Tensor:
    """CSR rows (device resident) -> binned int64 tokens [B, n_genes]; rows: int64 [B] row numbers or None for
    rows 0..batch-1.  err_flag: device int32 scalar the caller owns (bits are OR-ed in, never cleared here)."""
    _req(indptr, torch.int64, "indptr")
    _req(indices, torch.int32, "indices")
    _req(values, F32, "values")
    _req(err_flag, torch.int32, "err_flag")
    if rows is not None:
        _req(rows, torch.int64, "rows")
        batch = rows.numel()
    n_rows = indptr.numel() - 1
    tokens = torch.empty((batch, n_genes), dtype=torch.int64, device=values.device)
    if batch == 0:
        return tokens
    check(lib.scm_bin_csr(_p(indptr), _p(indices), _p(values), _p(rows), n_rows, batch, n_genes, rule, limit,
                          _p(tokens), _p(err_flag), _stream()), "scm_bin_csr")
    return tokens


def bin_dense(x: torch.Tensor, rule: int, limit: int) -> torch.Tensor:
    _req(x, F32, "x")
    tokens = torch.empty(x.shape, dtype=torch.int64, device=x.device)
    if x.numel():
        check(lib.scm_bin_dense(_p(x), x.numel(), rule, limit, _p(tokens), _stream()), "scm_bin_dense")
    return tokens


def orthogonal_omega(gauss_blocks: torch.Tensor, gauss_mult: Optional[torch.Tensor], nb_rows: int,
                     scaling: int = 0) -> torch.Tensor:
    """gauss_blocks f32 [nmat, ceil(nb_rows/64), 64, 64] (+ gauss_mult f32 [nmat, nb_rows, 64] for scaling 0)
    -> projection matrices f32 [nmat, nb_rows, 64]."""
    _req(gauss_blocks, F32, "gauss_blocks")
    nmat = gauss_blocks.shape[0]
    nblk = (nb_rows + DIM_HEAD - 1) // DIM_HEAD
    if tuple(gauss_blocks.shape) != (nmat, nblk, DIM_HEAD, DIM_HEAD):
        raise ValueError(f"gauss_blocks: expected [{nmat},{nblk},64,64], got {tuple(gauss_blocks.shape)}")
    if scaling == 0:
        _req(gauss_mult, F32, "gauss_mult")
        if tuple(gauss_mult.shape) != (nmat, nb_rows, DIM_HEAD):
            raise ValueError(f"gauss_mult: expected [{nmat},{nb_rows},64], got {tuple(gauss_mult.shape)}")
    omega = torch.empty((nmat, nb_rows, DIM_HEAD), dtype=F32, device=gauss_blocks.device)
    check(lib.scm_orthogonal_omega(_p(gauss_blocks), _p(gauss_mult) if scaling == 0 else None, nmat, nb_rows, scaling,
                                   _p(omega), _stream()), "scm_orthogonal_omega")
    return omega


# ----------------------------------------------------------------------------- per-kernel timing
def trace_kernel_names():
    return [lib.scm_trace_kernel_name(i).decode() for i in range(lib.scm_trace_kernel_count())]


def trace_enable(which):
    """which: None (off), 'all', or a kernel name from trace_kernel_names()."""
    names = trace_kernel_names()
    kid = -1 if which is None else (len(names) if which == "all" else names.index(which))
    check(lib.scm_trace_enable(kid), "scm_trace_enable")


def trace_collect():
    """-> {name: (milliseconds, algorithmic_flops, launches, algorithmic_bytes)} for kernels with launches;
    clears the log."""
    n = lib.scm_trace_kernel_count()
    ms, fl, by, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)()
    check(lib.scm_trace_collect(ms, fl, by, cnt), "scm_trace_collect")
    names = trace_kernel_names()
    return {names[i]: (ms[i], fl[i], cnt[i], by[i]) for i in range(n) if cnt[i] > 0}
