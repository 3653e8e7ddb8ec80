"""Thin tensor-level wrappers over the C-ABI building blocks (device tensors in/out).

PyTorch is plumbing here (device memory + the current HIP stream); all arithmetic runs in
the hand-written kernels of ``ganspace_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import math

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ganspace_amd ops need device tensors (no CPU fallback)")


def gram_accumulate(X, G=None, colsum=None, shift=None, precision="f32"):
    """``G += (X-shift)^T (X-shift)`` (float64 ``[d,d]``), ``colsum += sum(X-shift)`` (float64 ``[d]``)."""
    import torch
    lib = _lib.load()
    _need_cuda(X, G, colsum, shift)
    assert X.dtype == torch.float32 and X.dim() == 2 and X.stride(1) == 1
    rows, d = X.shape
    if G is None:
        G = torch.zeros((d, d), dtype=torch.float64, device=X.device)
    if colsum is None:
        colsum = torch.zeros(d, dtype=torch.float64, device=X.device)
    assert G.is_contiguous() and colsum.is_contiguous()
    if shift is not None:
        shift = shift.to(torch.float32).contiguous()
    _lib.check(lib.gs_gram_accumulate_prec(_p(X), rows, X.stride(0), d, _p(shift), _p(G), _p(colsum),
                                           _lib.PRECISIONS[precision], _lib.current_stream_ptr()))
    return G, colsum


def eigh_sym(A):
    """Eigen-decomposition of a symmetric PSD float64 matrix: returns ``(w, V, sweeps)`` with
    ``w`` descending and eigenvectors as the ROWS of ``V``."""
    import torch
    lib = _lib.load()
    _need_cuda(A)
    assert A.dtype == torch.float64 and A.dim() == 2 and A.shape[0] == A.shape[1]
    V = A.contiguous().clone()
    n = V.shape[0]
    w = torch.empty(n, dtype=torch.float64, device=A.device)
    sweeps = C.c_int(0)
    _lib.check(lib.gs_eigh_sym(_p(V), _p(w), n, C.cast(C.byref(sweeps), C.c_void_p), _lib.current_stream_ptr()))
    return w, V, sweeps.value


_mapping_ws = {}


def mapping_forward(z, weights, bias=None, lr_mul=0.01, slope=0.2, gain=math.sqrt(2.0), pixelnorm=True, out=None):
    """StyleGAN2 mapping network ``style(z)``: ``weights [L, dim, dim]``, ``bias [L, dim]``.  Row-wise - every output row is a
    function of its input row alone, bit for bit whatever the call length - so callers push many mini-batches through one
    call: at 80 000 rows the per-layer kernel runs at 0.76 of the f32-MFMA peak (two workgroups per CU), at one 10 000-row
    mini-batch at 0.62 (252 tiles on 256 CUs).  ``out`` may be a row slice of a larger array (the resident latents)."""
    import torch
    lib = _lib.load()
    _need_cuda(z, weights, bias, out)
    z = z.to(torch.float32).contiguous()
    weights = weights.to(torch.float32).contiguous()
    L, dim, dim2 = weights.shape
    assert dim == dim2 and z.shape[1] == dim
    if bias is not None:
        bias = bias.to(torch.float32).contiguous()
    w = out if out is not None else torch.empty_like(z)
    assert w.dtype == torch.float32 and w.shape == z.shape and w.is_contiguous() and w.data_ptr() != z.data_ptr()
    # ping-pong partner of `w`: one scratch per (device, stream), grown on demand - consecutive calls on a stream are ordered
    key = (z.device, torch.cuda.current_stream(z.device).cuda_stream)
    ws = _mapping_ws.get(key)
    if ws is None or ws.numel() < z.numel():
        ws = torch.empty(max(z.numel(), 4), dtype=torch.float32, device=z.device)
        _mapping_ws[key] = ws
    _lib.check(lib.gs_mapping_forward(_p(z), _p(w), _p(ws), _p(weights), _p(bias), L, dim,
                                      lr_mul / math.sqrt(dim), lr_mul, slope, gain, 1 if pixelnorm else 0,
                                      z.shape[0], _lib.current_stream_ptr()))
    return w


def linear_forward(x, weight, bias=None):
    """``torch.nn.functional.linear`` on the f32 matrix cores."""
    import torch
    lib = _lib.load()
    _need_cuda(x, weight, bias)
    x = x.to(torch.float32).contiguous()
    weight = weight.to(torch.float32).contiguous()
    if bias is not None:
        bias = bias.to(torch.float32).contiguous()
    out_f, in_f = weight.shape
    assert x.shape[1] == in_f
    y = torch.empty((x.shape[0], out_f), dtype=torch.float32, device=x.device)
    _lib.check(lib.gs_linear_forward(_p(x), _p(weight), _p(bias), _p(y), x.shape[0], in_f, out_f,
                                     _lib.current_stream_ptr()))
    return y


def _blocked_empty(rows, cols, device):
    import torch
    nb = C.c_int64(0)
    _lib.check(_lib.load().gs_blocked_nbytes(int(rows), int(cols), C.byref(nb)))
    return torch.empty(nb.value // 4, dtype=torch.float32, device=device)


def block_rows(mat):
    """Row-major ``[rows, cols]`` float32 matrix -> the panel-blocked operand of :func:`gemm_blocked_nt`."""
    import torch
    lib = _lib.load()
    _need_cuda(mat)
    mat = mat.to(torch.float32).contiguous()
    rows, cols = mat.shape
    dst = _blocked_empty(rows, cols, mat.device)
    _lib.check(lib.gs_block_rows(_p(mat), rows, cols, mat.stride(0), _p(dst), _lib.current_stream_ptr()))
    return dst


def im2col3x3_blocked(x_nhwc):
    """3 x 3 patches (zero padding 1) of a contiguous NHWC float32 tensor as the blocked ``[B*H*W, 9*C]`` operand."""
    import torch
    lib = _lib.load()
    _need_cuda(x_nhwc)
    assert x_nhwc.dtype == torch.float32 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    b, h, w, c = x_nhwc.shape
    dst = _blocked_empty(b * h * w, 9 * c, x_nhwc.device)
    _lib.check(lib.gs_im2col3x3_blocked(_p(x_nhwc), b, h, w, c, _p(dst), _lib.current_stream_ptr()))
    return dst


def gemm_blocked_nt(a_blocked, rows_a, b_blocked, rows_b, cols):
    """``A @ B.T`` (``[rows_a, rows_b]`` float32, row-major) from two panel-blocked operands with ``cols`` columns."""
    import torch
    lib = _lib.load()
    _need_cuda(a_blocked, b_blocked)
    out = torch.empty((rows_a, rows_b), dtype=torch.float32, device=a_blocked.device)
    _lib.check(lib.gs_gemm_blocked_nt(_p(a_blocked), rows_a, _p(b_blocked), rows_b, cols, _p(out), rows_b,
                                      _lib.current_stream_ptr()))
    return out


def modconv3x3_patches(x_nhwc, chan_scale=None, upsample=False):
    """Blocked 3 x 3 patch matrix of ``upsample2x(x * chan_scale[:, None, None, :])`` (``chan_scale`` [B, C] or None; the
    bilinear upsampling of ``F.interpolate(scale_factor=2, mode="bilinear", align_corners=False)``) without materialising the
    scaled or the upsampled tensor.  Rows = the ``B * H' * W'`` output pixels, columns ``(kh, kw, c)``."""
    import torch
    lib = _lib.load()
    _need_cuda(x_nhwc, chan_scale)
    assert x_nhwc.dtype == torch.float32 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    b, h, w, c = x_nhwc.shape
    if chan_scale is not None:
        chan_scale = chan_scale.to(torch.float32).contiguous()
        assert chan_scale.shape == (b, c)
    f = 2 if upsample else 1
    dst = _blocked_empty(b * h * f * w * f, 9 * c, x_nhwc.device)
    _lib.check(lib.gs_modconv3x3_patches(_p(x_nhwc), b, h, w, c, _p(chan_scale), 1 if upsample else 0, _p(dst),
                                         _lib.current_stream_ptr()))
    return dst


def gemm_blocked_nt_styled(a_blocked, rows_a, b_blocked, rows_b, cols, group_rows, group_colscale=None, row_add=None,
                           row_add_weight=0.0, bias=None, slope=0.2, gain=math.sqrt(2.0), act=True):
    """:func:`gemm_blocked_nt` with the StyledConv epilogue fused into the store:
    ``gain * lrelu(acc * group_colscale[row // group_rows, col] + row_add_weight * row_add[row % group_rows] + bias[col])``."""
    import torch
    lib = _lib.load()
    _need_cuda(a_blocked, b_blocked, group_colscale, row_add, bias)
    for t in (group_colscale, row_add, bias):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    assert group_colscale is None or group_colscale.numel() == -(-rows_a // group_rows) * rows_b
    assert row_add is None or row_add.numel() == group_rows
    assert bias is None or bias.numel() == rows_b
    out = torch.empty((rows_a, rows_b), dtype=torch.float32, device=a_blocked.device)
    _lib.check(lib.gs_gemm_blocked_nt_styled(_p(a_blocked), rows_a, _p(b_blocked), rows_b, cols, _p(out), rows_b, int(group_rows),
                                             _p(group_colscale), _p(row_add), float(row_add_weight), _p(bias), float(slope),
                                             float(gain), 1 if act else 0, _lib.current_stream_ptr()))
    return out


_project_scratch = {}


def project_rows(x, dirs, shift=None, colscale=None, out=None):
    """``((x - shift) @ dirs.T) * colscale`` for a few directions (``dirs`` [k, d], k small): the projection step of
    the regression (decomposition.py:110-118).  ``out`` may be a column slice ``buf[r0:r1, :k]`` of a wider row-major
    float32 buffer - the result is written in place, no intermediate copy of the batch or of the coordinates."""
    import torch
    lib = _lib.load()
    _need_cuda(x, dirs, shift, colscale, out)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    dirs = dirs.to(torch.float32).contiguous()
    k, d = dirs.shape
    assert x.shape[1] == d
    rows = x.shape[0]
    if shift is not None:
        shift = shift.to(torch.float32).contiguous().reshape(-1)
        assert shift.numel() == d
    if colscale is not None:
        colscale = colscale.to(torch.float32).contiguous().reshape(-1)
        assert colscale.numel() == k
    return _project_launch(x, _p(dirs), k, _p(shift), _p(colscale), out)


def _project_launch(x, dirs_ptr, k, shift_ptr, colscale_ptr, out):
    import torch
    lib = _lib.load()
    rows, d = x.shape
    if out is None:
        out = torch.empty((rows, k), dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.shape == (rows, k) and out.stride(1) == 1
    nb = C.c_int64(0)
    _lib.check(lib.gs_project_rows_nbytes(rows, k, d, C.byref(nb)))
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
    scratch = _project_scratch.get(key)
    if scratch is None or scratch.numel() < nb.value:
        scratch = torch.empty(nb.value, dtype=torch.uint8, device=x.device)
        _project_scratch[key] = scratch
    _lib.check(lib.gs_project_rows(_p(x), x.stride(0), rows, d, shift_ptr, dirs_ptr, k, colscale_ptr, _p(out),
                                   out.stride(0) if rows > 1 else max(out.stride(0), k), _p(scratch), scratch.numel(),
                                   _lib.current_stream_ptr()))
    return out


def project_rows_ptr(x, dirs_ptr, k, shift_ptr=None, out=None):
    """:func:`project_rows` with the directions (float32 ``[k, d]``, contiguous) and the shift (float32 ``[d]``) given as
    raw device pointers - the arrays ``gs_ipca_components_device`` hands out (``IncrementalPCA.transform``)."""
    import torch
    _need_cuda(x, out)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 4 == 0
    return _project_launch(x, dirs_ptr, int(k), shift_ptr if shift_ptr is not None else C.c_void_p(0), C.c_void_p(0), out)


def eigh_topk(A, k, V0=None):
    """Leading ``k`` eigenpairs of a symmetric PSD float64 matrix (Chebyshev-filtered subspace iteration):
    returns ``(w [k] descending, V [k, n] eigenvectors as rows, info)`` with
    ``info = dict(products, converged, sweeps, subspace)``."""
    import torch
    lib = _lib.load()
    _need_cuda(A, V0)
    assert A.dtype == torch.float64 and A.dim() == 2 and A.shape[0] == A.shape[1]
    A = A.contiguous()
    n = A.shape[0]
    V = torch.empty((k, n), dtype=torch.float64, device=A.device)
    w = torch.empty(k, dtype=torch.float64, device=A.device)
    k0 = 0
    if V0 is not None:
        V0 = V0.to(torch.float64).contiguous()
        k0 = V0.shape[0]
    info = (C.c_int * 4)()
    _lib.check(lib.gs_eigh_topk(_p(A), n, k, _p(V0), k0, _p(V), _p(w), C.cast(info, C.c_void_p),
                                _lib.current_stream_ptr()))
    return w, V, dict(products=info[0], converged=bool(info[1]), sweeps=info[2], subspace=info[3])


def cholqr(Y):
    """``Q = orth(Y)`` by CholeskyQR on the device (``Y``: ``[n, p]`` float64, p <= 128): ``(Q, rdiag)`` with
    ``rdiag`` the diagonal of the Cholesky factor of ``Y^T Y`` (0 marks a numerically dependent column)."""
    import torch
    lib = _lib.load()
    _need_cuda(Y)
    assert Y.dtype == torch.float64 and Y.dim() == 2
    Y = Y.contiguous()
    n, p = Y.shape
    Q = torch.empty_like(Y)
    rdiag = torch.empty(p, dtype=torch.float64, device=Y.device)
    _lib.check(lib.gs_cholqr(_p(Y), n, p, _p(Q), _p(rdiag), _lib.current_stream_ptr()))
    return Q, rdiag


def jacobi_small(B):
    """Single-workgroup eigensolver for a symmetric ``p x p`` float64 matrix (p % 8 == 0, p <= 128):
    ``(theta [p] descending, U [p, p] eigenvectors as columns, sweeps, limit_hit)``."""
    import torch
    lib = _lib.load()
    _need_cuda(B)
    assert B.dtype == torch.float64 and B.dim() == 2 and B.shape[0] == B.shape[1]
    B = B.contiguous()
    p = B.shape[0]
    U = torch.empty_like(B)
    theta = torch.empty(p, dtype=torch.float64, device=B.device)
    info = (C.c_int * 2)()
    _lib.check(lib.gs_jacobi_small(_p(B), p, _p(U), _p(theta), C.cast(info, C.c_void_p), _lib.current_stream_ptr()))
    return theta, U, info[0], bool(info[1])


def gemm_f64(A, B, alpha=1.0, beta=0.0, C=None, coef=None, E1=None, E2=None):
    """``C = alpha A @ B + beta C`` (or ``coef[0] A @ B + coef[1] E1 + coef[2] E2``) in float64 on the f64 matrix
    pipe; ``A`` / ``B`` may be arbitrary strided 2-D views (transposes are free)."""
    import torch
    lib = _lib.load()
    _need_cuda(A, B, C, coef, E1, E2)
    assert A.dtype == torch.float64 and B.dtype == torch.float64 and A.dim() == 2 and B.dim() == 2
    M, K = A.shape
    K2, N = B.shape
    assert K == K2
    if C is None:
        C = torch.zeros((M, N), dtype=torch.float64, device=A.device)
    assert C.stride(1) == 1 and C.shape == (M, N)
    for E in (E1, E2):
        assert E is None or (E.shape == C.shape and E.stride() == C.stride())
    _lib.check(lib.gs_gemm_f64(M, N, K, _p(A), A.stride(0), A.stride(1), _p(B), B.stride(0), B.stride(1), _p(C),
                               C.stride(0), float(alpha), float(beta), _p(coef), _p(E1), _p(E2),
                               _lib.current_stream_ptr()))
    return C


def eig_tridiag(B):
    """Eigen-decomposition of a symmetric ``p x p`` float64 matrix (p % 4 == 0, 8 <= p <= 128) by tridiagonalisation +
    bisection + twisted factorisations: ``(theta [p] descending, U [p, p] eigenvectors as columns, status)`` with status
    0 = fine, 2 = clustered eigenvalues (use ``jacobi_small``), 4 = non-finite."""
    import torch
    lib = _lib.load()
    _need_cuda(B)
    assert B.dtype == torch.float64 and B.dim() == 2 and B.shape[0] == B.shape[1]
    B = B.contiguous()
    p = B.shape[0]
    U = torch.empty_like(B)
    theta = torch.empty(p, dtype=torch.float64, device=B.device)
    info = (C.c_int * 2)()
    _lib.check(lib.gs_eig_tridiag(_p(B), p, _p(U), _p(theta), C.cast(info, C.c_void_p), _lib.current_stream_ptr()))
    return theta, U, info[1]
