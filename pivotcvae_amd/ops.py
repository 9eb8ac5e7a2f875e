"""Host-side operators: thin autograd wrappers over the C ABI (include/pcvae.h).

Every function takes ROCm tensors, marshals raw pointers + the current HIP stream into
libpcvae_hip.so and returns tensors allocated by torch's caching allocator.  Backward passes are
hand-written (they call the backward kernels); nothing here falls back to eager PyTorch math.
"""
import ctypes
import os

import torch

from . import _hip
from ._hip import (ACT_LEAKY, ACT_NONE, ACT_RELU, GEMM_DW, GEMM_DX, GEMM_DX_ACC, GEMM_FWD, GEMM_GROUP_MAX, GEMM_X3, GEMM_X6, PREC_BF16, PREC_BF16X3, PREC_BF16X6,
                   PREC_F32, PREC_SCREENED, GemmDesc, check, lib, ptr, require_device, stream)

F32 = torch.float32


def _c2d(t):
    """2-D fp32 view with contiguous rows (copy only if the layout forces it)."""
    if t.dtype != F32:
        raise TypeError(f"expected float32, got {t.dtype}")
    if t.dim() != 2:
        raise ValueError(f"expected a 2-D tensor, got {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ------------------------------------------------------------------------------ K1 / K2 / concat
def gather_rows(table, idx, out=None, group=1):
    """out[i // group, (i % group) * D : ...] = table[idx[i]]   (nn.Embedding lookup, frozen table).

    ``out`` may be a column window of a wider buffer (the torch.cat of the reference disappears).
    """
    require_device(table, idx, out)
    idx = idx.reshape(-1)
    if idx.dtype != torch.int64:
        idx = idx.to(torch.int64)
    idx = idx.contiguous()
    n, D = idx.numel(), table.shape[1]
    if n % group:
        raise ValueError("gather_rows: index count not a multiple of group")
    if out is None:
        out = torch.empty(n // group, group * D, dtype=F32, device=table.device)
    check(lib().pcvae_gather_rows(ptr(table, F32), table.shape[0], D, ptr(idx), n, group, ptr(out, F32), _ld(out),
                                  stream()), "gather_rows")
    return out


def condition(r, S, out=None):
    """one-hot of the click count, [B, S+1] (models/cvae.py:85-92).  r is [B, any width]: the reference's in-loop
    evaluation passes a 5-column context whatever the slate size is (train_generative.py:179)."""
    require_device(r, out)
    if r.dim() != 2:
        raise RuntimeError(f"condition: r must be [B, columns], got {tuple(r.shape)}")
    r = r.to(F32).contiguous()
    B, ncols = r.shape
    if out is None:
        out = torch.empty(B, S + 1, dtype=F32, device=r.device)
    check(lib().pcvae_condition(ptr(r, F32), B, ncols, S, ptr(out, F32), _ld(out), stream()), "condition")
    return out


def copy2d(src, dst):
    require_device(src, dst)
    src = _c2d(src)
    check(lib().pcvae_copy2d(ptr(src, F32), _ld(src), ptr(dst, F32), _ld(dst), src.shape[0], src.shape[1], stream()),
          "copy2d")
    return dst


class _Concat(torch.autograd.Function):
    """torch.cat(parts, 1) as strided copies into one buffer; backward hands out column windows."""

    @staticmethod
    def forward(ctx, *parts):
        require_device(*parts)
        B = parts[0].shape[0]
        widths = [p.shape[1] for p in parts]
        out = torch.empty(B, sum(widths), dtype=F32, device=parts[0].device)
        c = 0
        for i in range(0, len(parts), 4):   # up to four parts per launch
            grp = [_c2d(p) for p in parts[i:i + 4]]
            args = []
            for k in range(4):
                args += [ptr(grp[k], F32), _ld(grp[k]), grp[k].shape[1]] if k < len(grp) else [None, 0, 0]
            w = sum(g.shape[1] for g in grp)
            dst = out[:, c:c + w]
            check(lib().pcvae_concat(*args, ptr(dst, F32), _ld(dst), B, stream()), "concat")
            c += w
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, g):
        outs, c = [], 0
        for i, w in enumerate(ctx.widths):
            outs.append(g[:, c:c + w] if ctx.needs_input_grad[i] else None)
            c += w
        return tuple(outs)


def concat(parts):
    return _Concat.apply(*parts)


# ------------------------------------------------------------------------------------------- K3
# bench.py sets this to (begin() -> token, end(token, flops, launches)) to bracket the MLP GEMM launches with HIP events on the
# launch stream (MFMA utilisation of the MLP stacks); None in normal operation.  A stack's forward (or backward) pass is a run of
# dependent GEMM launches with nothing between them: inside gemm_span() they are timed as ONE interval, launch gaps included.
GEMM_TIMING = None
_span = None   # [flops, launches, token] of the open span


class gemm_span:
    def __enter__(self):
        global _span
        self.outer = _span
        if GEMM_TIMING is not None and _span is None:
            _span = [0.0, 0, None]
        return self

    def __exit__(self, *exc):
        global _span
        if self.outer is None and _span is not None:
            flops, launches, tok = _span
            _span = None
            if tok is not None:
                GEMM_TIMING[1](tok, flops, launches)
        return False


def _timed_gemm(flops, launch):
    t = GEMM_TIMING
    if t is None:
        return launch()
    if _span is not None:
        if _span[2] is None:
            _span[2] = t[0]()
        _span[0] += flops
        _span[1] += 1
        return launch()
    tok = t[0]()
    launch()
    t[1](tok, flops, 1)


# Arithmetic of the MLP GEMMs (csrc/gemm_f32.hip):
#   "f32"     exact fp32 MFMA (v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain);
#   "bf16x3"  operands split into bf16 hi + lo in registers, three bf16 MFMAs per product, fp32 accumulate - 16-bit-mantissa operands,
#             2^-18 relative per product: a stated-tolerance fast path, fp32-equivalent at the GEMM tests' tolerances;
#   "bf16x6"  (round 6) every fp32 operand as THREE bf16 components whose sum is the fp32 value exactly, six bf16 MFMAs per product
#             (the dropped pairs are <= 2^-25 relative: below the rounding of an fp32 product), fp32 accumulate - the reference's fp32
#             arithmetic on the bf16 matrix cores, the catalog kernel's bf16x6 applied to K3.
# A model sets it around its training loss (BaseCVAE.set_mlp_precision); an autograd node remembers the arithmetic of its forward
# and runs its backward in the same one.
MLP_PRECISIONS = ("f32", "bf16x3", "bf16x6")
MLP_ARITHMETIC = {"f32": "f32 (v_mfma_f32_32x32x2_f32)",
                  "bf16x3": "bf16x3: operands as bf16 hi+lo in registers, 3 bf16 MFMAs per product, fp32 accumulate (train step only; "
                            "generation: exact f32)",
                  "bf16x6": "bf16x6: fp32 operands as 3 bf16 components in registers, 6 bf16 MFMAs per product, fp32 accumulate - "
                            "fp32-exact products (train step only; generation: exact f32)"}
_MLP_MODE = 0   # 0 f32, 1 bf16x3, 2 bf16x6 (the flag OR-ed into a GEMM descriptor's kind)
_MODE_OF = {"f32": 0, "fp32": 0, "bf16x3": 1, "bf16x6": 2, False: 0, True: 1, None: 0, 0: 0, 1: 1, 2: 2}


def default_mlp_precision(catalog_dtype):
    """the MLP arithmetic that goes with a catalog arithmetic: bf16x3 where the catalog contraction is bf16x3 / bf16 (a stated-
    tolerance step end to end), bf16x6 where it is bf16x6 (fp32-exact products on the bf16 matrix cores end to end), exact f32 else"""
    return {"bf16x3": "bf16x3", "bf16": "bf16x3", "bf16x6": "bf16x6"}.get(catalog_dtype, "f32")


class mlp_arith:
    """``with mlp_arith("bf16x6"):`` the GEMMs launched inside run in that arithmetic ("f32" / "bf16x3" / "bf16x6"; the round-3
    booleans still mean f32 / bf16x3)"""

    def __init__(self, mode):
        self.mode = _MODE_OF[mode]

    def __enter__(self):
        global _MLP_MODE
        self.prev, _MLP_MODE = _MLP_MODE, self.mode
        return self

    def __exit__(self, *exc):
        global _MLP_MODE
        _MLP_MODE = self.prev
        return False


def _one(build):
    """a single GEMM through the grouped entry point (the only one that carries the bf16x3 flag)"""
    grp = GemmGroup()
    out = build(grp)
    grp.launch()
    return out


def linear_fwd_raw(x, W, b, act, out=None):
    if _MLP_MODE:
        return _one(lambda g: g.fwd(x, W, b, act, out=out))
    x = _c2d(x)
    M, K = x.shape
    N = W.shape[0]
    if W.shape[1] != K:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({M}x{K} and {W.shape[1]}x{N})")
    if out is None:
        out = torch.empty(M, N, dtype=F32, device=x.device)
    _timed_gemm(2.0 * M * N * K, lambda: check(
        lib().pcvae_linear_fwd(ptr(x, F32), _ld(x), ptr(W, F32), _ld(W), ptr(b, F32) if b is not None else None,
                               ptr(out, F32), _ld(out), M, N, K, act, stream()), "linear_fwd"))
    return out


def linear_bwd_input_raw(gy, W, xact=None, out=None):
    if _MLP_MODE:
        return _one(lambda g: g.dx(gy, W, xact=xact, out=out))
    gy = _c2d(gy)
    M, N = gy.shape
    K = W.shape[1]
    if out is None:
        out = torch.empty(M, K, dtype=F32, device=gy.device)
    _timed_gemm(2.0 * M * N * K, lambda: check(
        lib().pcvae_linear_bwd_input(ptr(gy, F32), _ld(gy), ptr(W, F32), _ld(W),
                                     ptr(xact, F32) if xact is not None else None,
                                     _ld(xact) if xact is not None else 0, ptr(out, F32), _ld(out), M, N, K,
                                     stream()), "linear_bwd_input"))
    return out


def linear_bwd_weight_raw(gy, x, dW, db):
    """dW += gy^T x ; db += colsum(gy)  (accumulating into the given buffers)."""
    grp = GemmGroup()
    grp.dw(gy, x, dW, db)
    grp.launch()


class GemmGroup:
    """Independent layer GEMMs collected into ONE launch (pcvae_linear_group): a layer's weight- and input-gradient, the same
    layer of two stacks that do not feed each other.  The methods mirror linear_fwd_raw / linear_bwd_input_raw /
    linear_bwd_weight_raw; nothing runs before launch()."""

    def __init__(self):
        self.descs, self.keep, self.flops = [], [], 0.0
        self.mode = _MLP_MODE

    def _add(self, kind, act, a, b, c, aux, aux_out, M, N, K):
        d = GemmDesc(kind | (0, GEMM_X3, GEMM_X6)[self.mode], act, a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), c.data_ptr(), _ld(c),
                     aux.data_ptr() if aux is not None else None, (_ld(aux) if aux is not None and aux.dim() == 2 else 0),
                     aux_out.data_ptr() if aux_out is not None else None, M, N, K)
        for x in (a, b, c, aux, aux_out):
            if x is not None and x.dtype != F32:
                raise TypeError(f"expected {F32}, got {x.dtype}")
        self.descs.append(d)
        self.keep += [a, b, c, aux, aux_out]
        self.flops += 2.0 * M * N * K

    @property
    def x3(self):   # (round-3 name) does this group run on the bf16 matrix cores (bf16x3 or bf16x6)?
        return self.mode != 0

    def fwd(self, x, W, b, act, out=None):
        x = _c2d(x)
        M, K = x.shape
        N = W.shape[0]
        if W.shape[1] != K:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({M}x{K} and {W.shape[1]}x{N})")
        if out is None:
            out = torch.empty(M, N, dtype=F32, device=x.device)
        self._add(GEMM_FWD, act, x, W, out, b, None, M, N, K)
        return out

    def dx(self, gy, W, xact=None, out=None, accumulate=False, cols=None):
        """out[M, cols] = gy @ W[:, :cols] (* LeakyReLU'(xact)); cols = None: every input column"""
        gy = _c2d(gy)
        M, N = gy.shape
        K = W.shape[1] if cols is None else int(cols)
        if out is None:
            out = torch.empty(M, K, dtype=F32, device=gy.device)
        self._add(GEMM_DX_ACC if accumulate else GEMM_DX, 0, gy, W, out, xact, None, M, N, K)
        return out

    def dw(self, gy, x, dW, db):
        gy, x = _c2d(gy), _c2d(x)
        M, N = gy.shape
        self._add(GEMM_DW, 0, gy, x, dW, None, db, M, N, x.shape[1])

    def launch(self):
        for i in range(0, len(self.descs), GEMM_GROUP_MAX):
            part = self.descs[i:i + GEMM_GROUP_MAX]
            arr = (GemmDesc * len(part))(*part)
            flops = sum(2.0 * d.M * d.N * d.K for d in part)
            # weight gradients combine their batch splits through a scratch buffer, in split order (no fp32 atomics: those lose
            # updates between XCDs - csrc/common.h; bitwise reproducible)
            nbytes = lib().pcvae_linear_group_ws_bytes(arr, len(part))
            ws = _workspace(self.keep[0].device, nbytes, tag="gemm", zero=True) if nbytes else None
            _timed_gemm(flops, lambda: check(lib().pcvae_linear_group(arr, len(part), ptr(ws), nbytes, stream()), "linear_group"))
        self.descs, self.keep = [], []


def leaky_bwd_(g, y):
    check(lib().pcvae_leaky_bwd(ptr(g, F32), _ld(g), ptr(y, F32), _ld(y), g.shape[0], g.shape[1], stream()),
          "leaky_bwd")
    return g


class _MLP(torch.autograd.Function):
    """A stack of Linear(+LeakyReLU) layers as one autograd node.

    forward keeps the activated outputs; backward walks the stack once: LeakyReLU' of an inner layer is
    fused into the epilogue of the input-gradient GEMM of the layer above it.  When a parameter already owns a
    gradient buffer (FlatAdam attaches views of its flat gradient buffer), the weight-gradient GEMM accumulates
    straight into it - the kernel is an accumulating one anyway - instead of materialising a temporary that
    autograd would then have to add (saves a fill + an add launch per parameter tensor).
    """

    @staticmethod
    def forward(ctx, x, last_linear, *params):
        ctx.x3 = _MLP_MODE
        with gemm_span():
            return _MLP._forward_impl(ctx, x, last_linear, *params)

    @staticmethod
    def _forward_impl(ctx, x, last_linear, *params):
        require_device(x, *params)
        n = len(params) // 2
        x = _c2d(x)
        acts = [x]
        h = x
        for i in range(n):
            act = ACT_NONE if (last_linear and i == n - 1) else ACT_LEAKY
            h = linear_fwd_raw(h, params[2 * i], params[2 * i + 1], act)
            acts.append(h)
        ctx.last_linear = last_linear
        ctx.n = n
        # gradient buffers to accumulate into directly (leaf parameters with a pre-attached, same-shape .grad)
        ctx.direct = [p.grad if (p.is_leaf and p.requires_grad and p.grad is not None and p.grad.is_cuda
                                 and p.grad.shape == p.shape and p.grad.is_contiguous()) else None for p in params]
        ctx.save_for_backward(*acts, *params)
        return h

    @staticmethod
    def backward(ctx, g):
        with gemm_span(), mlp_arith(ctx.x3):
            return _MLP._backward_impl(ctx, g)

    @staticmethod
    def _backward_impl(ctx, g):
        n = ctx.n
        saved = ctx.saved_tensors
        acts, params = saved[: n + 1], saved[n + 1:]
        g = _c2d(g)
        if not ctx.last_linear:  # the top layer is activated: apply its LeakyReLU' explicitly
            g = leaky_bwd_(g.clone(), acts[n])
        grads = [None] * (2 * n)
        dwg = GemmGroup()   # the weight gradients are off the dependency chain: ONE grouped launch behind the input-gradient chain
        for i in range(n - 1, -1, -1):
            W, b = params[2 * i], params[2 * i + 1]
            if ctx.needs_input_grad[2 + 2 * i] or ctx.needs_input_grad[3 + 2 * i]:
                dW, db = ctx.direct[2 * i], ctx.direct[2 * i + 1]
                if dW is None or db is None:   # else: accumulated in place, nothing to hand to autograd
                    dW, db = torch.zeros_like(W), torch.zeros_like(b)
                    grads[2 * i], grads[2 * i + 1] = dW, db
                dwg.dw(g, acts[i], dW, db)
            if i > 0:
                g = linear_bwd_input_raw(g, W, xact=acts[i])  # acts[i] is layer i-1's activated output
            elif ctx.needs_input_grad[0]:
                g = linear_bwd_input_raw(g, W, xact=None)
            else:
                g = None
        dwg.launch()
        return (g, None) + tuple(grads)


def linear_bwd_input_acc_raw(gy, W, xact, out):
    """out = (out + gy @ W) * LeakyReLU'(xact)   (xact may be None: no mask)."""
    if _MLP_MODE:
        return _one(lambda g: g.dx(gy, W, xact=xact, out=out, accumulate=True))
    gy = _c2d(gy)
    M, N = gy.shape
    K = W.shape[1]
    _timed_gemm(2.0 * M * N * K, lambda: check(
        lib().pcvae_linear_bwd_input_acc(ptr(gy, F32), _ld(gy), ptr(W, F32), _ld(W),
                                         ptr(xact, F32) if xact is not None else None,
                                         _ld(xact) if xact is not None else 0, ptr(out, F32), _ld(out), M, N, K,
                                         stream()), "linear_bwd_input_acc"))
    return out


class _MLPHeads(torch.autograd.Function):
    """A LeakyReLU trunk followed by TWO linear heads on its output (encoder -> mu / logvar, prior -> mu / logvar;
    models/pivotcvae.py:205-220,232-239) as one autograd node.  As separate nodes the two heads hand autograd two input
    gradients to add, and the trunk then copies the sum and masks it with LeakyReLU' - three launches that the second head's
    input-gradient GEMM absorbs here (pcvae_linear_bwd_input_acc: accumulate + mask in the epilogue)."""

    @staticmethod
    def forward(ctx, x, n_trunk, *params):
        ctx.x3 = _MLP_MODE
        with gemm_span():
            return _MLPHeads._forward_impl(ctx, x, n_trunk, *params)

    @staticmethod
    def _forward_impl(ctx, x, n_trunk, *params):
        require_device(x, *params)
        x = _c2d(x)
        acts = [x]
        h = x
        for i in range(n_trunk):
            h = linear_fwd_raw(h, params[2 * i], params[2 * i + 1], ACT_LEAKY)
            acts.append(h)
        Wa, ba, Wb, bb = params[2 * n_trunk:2 * n_trunk + 4]
        ya = linear_fwd_raw(h, Wa, ba, ACT_NONE)
        yb = linear_fwd_raw(h, Wb, bb, ACT_NONE)
        ctx.n = n_trunk
        ctx.direct = [p.grad if (p.is_leaf and p.requires_grad and p.grad is not None and p.grad.is_cuda
                                 and p.grad.shape == p.shape and p.grad.is_contiguous()) else None for p in params]
        ctx.save_for_backward(*acts, *params)
        return ya, yb

    @staticmethod
    def backward(ctx, ga, gb):
        with gemm_span(), mlp_arith(ctx.x3):
            return _MLPHeads._backward_impl(ctx, ga, gb)

    @staticmethod
    def _backward_impl(ctx, ga, gb):
        n = ctx.n
        saved = ctx.saved_tensors
        acts, params = saved[: n + 1], saved[n + 1:]
        grads = [None] * len(params)
        h = acts[n]

        def weight_grad(grp, k, g, a):   # parameter pair k (weight 2k, bias 2k + 1) from output gradient g and layer input a
            if not (ctx.needs_input_grad[2 + 2 * k] or ctx.needs_input_grad[3 + 2 * k]):
                return
            dW, db = ctx.direct[2 * k], ctx.direct[2 * k + 1]
            if dW is None or db is None:   # else: accumulated in place, nothing to hand to autograd
                dW, db = torch.zeros_like(params[2 * k]), torch.zeros_like(params[2 * k + 1])
                grads[2 * k], grads[2 * k + 1] = dW, db
            grp.dw(g, a, dW, db)

        ga, gb = _c2d(ga), _c2d(gb)
        dwg = GemmGroup()   # every weight gradient of the node: ONE grouped launch behind the input-gradient chain
        weight_grad(dwg, n, ga, h)
        weight_grad(dwg, n + 1, gb, h)
        if n == 0 and not ctx.needs_input_grad[0]:
            dwg.launch()
            return (None, None) + tuple(grads)
        # d h = ga Wa + gb Wb, masked with the trunk's top LeakyReLU' in the second GEMM's epilogue
        g = linear_bwd_input_raw(ga, params[2 * n], xact=None)
        g = linear_bwd_input_acc_raw(gb, params[2 * n + 2], h if n > 0 else None, g)
        for i in range(n - 1, -1, -1):
            weight_grad(dwg, i, g, acts[i])
            if i > 0:
                g = linear_bwd_input_raw(g, params[2 * i], xact=acts[i])
            elif ctx.needs_input_grad[0]:
                g = linear_bwd_input_raw(g, params[0], xact=None)
            else:
                g = None
        dwg.launch()
        return (g, None) + tuple(grads)


def mlp_heads(x, trunk, head_a, head_b):
    """trunk: list of (weight, bias) with LeakyReLU after every layer; head_a / head_b: (weight, bias), linear.
    -> (head_a(trunk(x)), head_b(trunk(x)))"""
    flat = []
    for W, b in list(trunk) + [head_a, head_b]:
        flat += [W, b]
    return _MLPHeads.apply(x, len(trunk), *flat)


# ---------------------------------------------------------------- fused training-path operators (fewer, larger launches)
# bench.py sets this to (begin() -> token, end(token, bytes_moved)) to bracket the launch with HIP events inside real train steps
ASSEMBLE_TIMING = None
ASSEMBLE_ROW_ALIGN = 1   # floats; 1 = packed rows (the product).  The alignment probe sets it (see assemble_inputs)


def assemble_inputs(E, U, s, r, u, Z):
    """condition + item / user / pivot gathers + the reference's concatenations in ONE launch (pcvae_assemble_inputs).
    -> (enc_in [B, S D + C (+D)], prior_in [B, C (+D)], scm_in [B, Z + C + D (+D)] with its z window unwritten, rx [B, S D] with
    slot 0 = the ground-truth pivot's row)."""
    require_device(E, U, s, r, u)
    B, S = s.shape
    D, C = E.shape[1], S + 1
    ud = 0 if U is None else D
    dev = E.device
    s = s.to(torch.int64).contiguous()
    r = r.to(F32).contiguous()
    uu = None if U is None else u.reshape(-1).to(torch.int64).contiguous()
    # Rows are PACKED (ld = width: 1419 / 139 / 283 floats at config 4, so rows start on arbitrary 4-byte boundaries).  Round 6
    # measured rows padded to 4 / 16 / 32 / 64 floats instead (tools/assemble_align_probe.py, profiles/r06_assemble_row_align_probe.txt):
    # 26.5 - 27.5 us against 26.7 us packed - partial cache lines are not what bounds this kernel.  ASSEMBLE_ROW_ALIGN > 1 is the
    # probe's knob only (the padded buffers are views, which _LatentPacked's in-place write of z cannot take).
    def rows(width):
        if ASSEMBLE_ROW_ALIGN <= 1:
            return torch.empty(B, width, dtype=F32, device=dev)
        ld = -(-width // ASSEMBLE_ROW_ALIGN) * ASSEMBLE_ROW_ALIGN
        return torch.empty(B, ld, dtype=F32, device=dev)[:, :width]

    enc_in = rows(S * D + C + ud)
    prior_in = rows(C + ud)
    scm_in = rows(Z + C + D + ud)
    rx = torch.empty(B, S * D, dtype=F32, device=dev)
    timing = ASSEMBLE_TIMING
    tok = timing[0]() if timing is not None else None
    check(lib().pcvae_assemble_inputs(ptr(E, F32), E.shape[0], ptr(U, F32) if U is not None else None,
                                      U.shape[0] if U is not None else 0, ptr(s), ptr(r, F32), ptr(uu), B, S, D, r.shape[1], Z,
                                      ptr(enc_in, F32), _ld(enc_in), ptr(prior_in, F32), _ld(prior_in), ptr(scm_in, F32), _ld(scm_in),
                                      ptr(rx, F32), _ld(rx), stream()), "assemble_inputs")
    if timing is not None:   # rows read once + the four outputs written + the int64 indices
        timing[1](tok, B * (4 * ((S + (U is not None)) * D + enc_in.shape[1] + prior_in.shape[1] + (scm_in.shape[1] - Z) + D)
                            + 8 * (S + (U is not None))))
    return enc_in, prior_in, scm_in, rx


def heads_adjacent(head_a, head_b):
    """can the two heads run as ONE N = 2 Z GEMM?  Their weights (and biases, and the gradient buffers attached to all four) must
    be back to back in memory - what FlatAdam arranges for the groups a model lists in flat_param_groups()."""
    Wa, ba, Wb, bb = head_a.weight, head_a.bias, head_b.weight, head_b.bias
    if Wa.shape != Wb.shape or ba.shape != bb.shape or not (Wa.is_contiguous() and Wb.is_contiguous()):
        return False

    def back_to_back(x, y):
        return x is not None and y is not None and x.is_cuda and x.is_contiguous() and y.is_contiguous() and \
            y.data_ptr() == x.data_ptr() + x.numel() * 4

    return back_to_back(Wa.data, Wb.data) and back_to_back(ba.data, bb.data) and back_to_back(Wa.grad, Wb.grad) and \
        back_to_back(ba.grad, bb.grad)


def _cat2(a, b):
    """[a ; b] as one view: a and b are back to back in memory (heads_adjacent)"""
    shape = (2 * a.shape[0],) + tuple(a.shape[1:])
    return torch.as_strided(a, shape, a.stride())


class _StacksPacked(torch.autograd.Function):
    """One or several INDEPENDENT stacks, each a LeakyReLU trunk + its two linear heads as ONE GEMM with N = 2 Z
    (models/pivotcvae.py:167-173, 232-239); the output of a stack is the packed [mu | logvar].  Layer i of every stack is one
    grouped launch (the encoder and the prior share only their inputs), and in backward so are a layer's weight- and
    input-gradient.  Needs heads_adjacent(); gradients accumulate straight into the flat gradient buffer."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        ctx.x3 = _MLP_MODE
        with gemm_span():
            return _StacksPacked._forward_impl(ctx, spec, *tensors)

    @staticmethod
    def _forward_impl(ctx, spec, *tensors):
        S = len(spec)
        require_device(*tensors)
        xs = [_c2d(x) for x in tensors[:S]]
        params, pos = [], S
        for n in spec:
            params.append(tensors[pos:pos + 2 * n + 4])
            pos += 2 * n + 4
        acts = [[x] for x in xs]
        ys = [None] * S
        for lvl in range(max(spec) + 1):
            grp = GemmGroup()
            for s, n in enumerate(spec):
                P = params[s]
                if lvl < n:
                    acts[s].append(grp.fwd(acts[s][-1], P[2 * lvl], P[2 * lvl + 1], ACT_LEAKY))
                elif lvl == n:
                    Wa, ba, Wb, bb = P[2 * n:2 * n + 4]
                    ys[s] = grp.fwd(acts[s][-1], _cat2(Wa, Wb), _cat2(ba, bb), ACT_NONE)
            grp.launch()
        ctx.spec = spec
        ctx.direct = [[q.grad for q in P] for P in params]   # views of the flat gradient buffer: accumulated into in place
        if any(g is None for D in ctx.direct for g in D):
            raise RuntimeError("mlp_heads_packed needs gradient buffers attached to every parameter (FlatAdam)")
        ctx.save_for_backward(*[a for A in acts for a in A], *[q for P in params for q in P])
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        with gemm_span(), mlp_arith(ctx.x3):
            return _StacksPacked._backward_impl(ctx, *gs)

    @staticmethod
    def _backward_impl(ctx, *gs):
        spec = ctx.spec
        S = len(spec)
        saved = list(ctx.saved_tensors)
        acts, pos = [], 0
        for n in spec:
            acts.append(saved[pos:pos + n + 1])
            pos += n + 1
        params = []
        for n in spec:
            params.append(saved[pos:pos + 2 * n + 4])
            pos += 2 * n + 4
        g = [_c2d(x) for x in gs]
        dwg = GemmGroup()   # the weight gradients of every layer of every stack: off the dependency chain, ONE grouped launch (six
        #                     problems per launch) behind the input-gradient chain - the reductions of their batch splits overlap
        for lvl in range(max(spec) + 1):   # level 0 = the heads, level j = trunk layer n - j
            grp = GemmGroup()
            for s, n in enumerate(spec):
                P, D, need_x = params[s], ctx.direct[s], ctx.needs_input_grad[1 + s]
                if lvl == 0:
                    Wa, ba, Wb, bb = P[2 * n:2 * n + 4]
                    dWa, dba, dWb, dbb = D[2 * n:2 * n + 4]
                    h = acts[s][n]
                    dwg.dw(g[s], h, _cat2(dWa, dWb), _cat2(dba, dbb))
                    g[s] = grp.dx(g[s], _cat2(Wa, Wb), xact=h if n > 0 else None) if (n > 0 or need_x) else None
                elif lvl <= n:
                    i = n - lvl
                    dwg.dw(g[s], acts[s][i], D[2 * i], D[2 * i + 1])
                    g[s] = grp.dx(g[s], P[2 * i], xact=acts[s][i] if i > 0 else None) if (i > 0 or need_x) else None
            if grp.descs:
                grp.launch()
        dwg.launch()
        return (None,) + tuple(g) + (None,) * (len(saved) - sum(n + 1 for n in spec))


def _stack_args(x, trunk, head_a, head_b):
    flat = []
    for W, b in list(trunk) + [head_a, head_b]:
        flat += [W, b]
    return x, flat


def mlp_heads_packed(x, trunk, head_a, head_b):
    """-> [head_a(trunk(x)) | head_b(trunk(x))] as one [B, 2 Z] tensor (heads_adjacent(head_a, head_b) must hold)"""
    x, flat = _stack_args(x, trunk, head_a, head_b)
    return _StacksPacked.apply((len(trunk),), x, *flat)[0]


def mlp_heads_packed_pair(stack_a, stack_b):
    """two independent stacks (x, trunk, head_a, head_b) layer by layer in grouped launches -> (y_a, y_b), each as mlp_heads_packed"""
    xa, fa = _stack_args(*stack_a)
    xb, fb = _stack_args(*stack_b)
    return _StacksPacked.apply((len(stack_a[1]), len(stack_b[1])), xa, xb, *fa, *fb)


class _LatentPacked(torch.autograd.Function):
    """reparametrize + KLD on packed head outputs; z is written into the first Z columns of the slate-completion input, which is
    returned (the reference's torch.cat([z, cond, pivot, user]) never happens)."""

    @staticmethod
    def forward(ctx, y_enc, y_prior, scm_in, eps, seed, offset, Z):
        require_device(y_enc, y_prior, scm_in, eps)
        y_enc, y_prior = _c2d(y_enc), _c2d(y_prior)
        B = y_enc.shape[0]
        if eps is not None:
            eps = eps.to(F32).contiguous()
        eps_used = torch.empty(B, Z, dtype=F32, device=y_enc.device)
        k = torch.empty((), dtype=F32, device=y_enc.device)
        if _ld(y_enc) != _ld(y_prior):
            raise RuntimeError("latent_packed: the two packed inputs must share a leading dimension")
        check(lib().pcvae_latent_fwd_packed(ptr(y_enc, F32), ptr(y_prior, F32), _ld(y_enc), ptr(eps, F32) if eps is not None else None,
                                            seed, offset, ptr(scm_in, F32), _ld(scm_in), ptr(eps_used, F32), ptr(k, F32), B, Z,
                                            stream()), "latent_fwd_packed")
        ctx.Z = Z
        ctx.save_for_backward(eps_used, y_enc, y_prior)
        ctx.mark_dirty(scm_in)
        ctx.mark_non_differentiable(eps_used)
        ctx.set_materialize_grads(False)
        return scm_in, eps_used, k

    @staticmethod
    def backward(ctx, g_scm, _geps, gk):
        eps, y_enc, y_prior = ctx.saved_tensors
        B, Z = eps.shape
        g_enc = torch.empty(B, 2 * Z, dtype=F32, device=eps.device)
        g_prior = torch.empty(B, 2 * Z, dtype=F32, device=eps.device)
        if g_scm is not None:
            g_scm = _c2d(g_scm)
        gk_ptr, gk_host = (ptr(gk.contiguous(), F32), 1.0) if gk is not None else (None, 0.0)
        check(lib().pcvae_latent_bwd_packed(ptr(g_scm, F32) if g_scm is not None else None, _ld(g_scm) if g_scm is not None else 0,
                                            ptr(eps, F32), ptr(y_enc, F32), ptr(y_prior, F32), _ld(y_enc), gk_ptr, gk_host,
                                            ptr(g_enc, F32), ptr(g_prior, F32), 2 * Z, B, Z, stream()), "latent_bwd_packed")
        return g_enc, g_prior, None, None, None, None, None


def latent_packed(y_enc, y_prior, scm_in, eps=None, seed=0, offset=0, Z=None):
    """-> (scm_in with z in its first Z columns, eps_used, kld)"""
    return _LatentPacked.apply(y_enc, y_prior, scm_in, eps, int(seed), int(offset), int(Z if Z is not None else y_enc.shape[1] // 2))


class _MLPInto(torch.autograd.Function):
    """_MLP (last layer linear) whose last layer writes into the column window [col0, col0 + out) of a prepared buffer, which is
    returned whole: the slate-completion stack fills slots 1.. of rx next to the pivot row (models/pivotcvae.py:222-226's
    reshape + cat never happens).  The columns in front of col0 are constants for autograd.  grad_cols: only the first
    grad_cols columns of x carry a gradient (x = [z | condition | pivot row | user row]: everything behind z comes from frozen
    tables) - the bottom layer's input-gradient GEMM computes just those, the rest of the returned gradient is UNWRITTEN memory
    that the producer of x (latent_packed) never reads."""

    @staticmethod
    def forward(ctx, x, out_buf, col0, grad_cols, *params):
        ctx.x3 = _MLP_MODE
        with gemm_span():
            return _MLPInto._forward_impl(ctx, x, out_buf, col0, grad_cols, *params)

    @staticmethod
    def _forward_impl(ctx, x, out_buf, col0, grad_cols, *params):
        require_device(x, out_buf, *params)
        n = len(params) // 2
        x = _c2d(x)
        acts = [x]
        h = x
        for i in range(n - 1):
            h = linear_fwd_raw(h, params[2 * i], params[2 * i + 1], ACT_LEAKY)
            acts.append(h)
        width = params[2 * n - 2].shape[0]
        linear_fwd_raw(h, params[2 * n - 2], params[2 * n - 1], ACT_NONE, out=out_buf[:, col0:col0 + width])
        ctx.n, ctx.col0, ctx.width, ctx.grad_cols = n, col0, width, grad_cols
        ctx.direct = [p.grad if (p.is_leaf and p.requires_grad and p.grad is not None and p.grad.is_cuda
                                 and p.grad.shape == p.shape and p.grad.is_contiguous()) else None for p in params]
        ctx.save_for_backward(*acts, *params)
        ctx.mark_dirty(out_buf)
        return out_buf

    @staticmethod
    def backward(ctx, g):
        with gemm_span(), mlp_arith(ctx.x3):
            return _MLPInto._backward_impl(ctx, g)

    @staticmethod
    def _backward_impl(ctx, g):
        n = ctx.n
        saved = ctx.saved_tensors
        acts, params = saved[:n], saved[n:]
        g = _c2d(g)[:, ctx.col0:ctx.col0 + ctx.width]
        grads = [None] * (2 * n)
        dwg = GemmGroup()   # the weight gradients are off the dependency chain: ONE grouped launch behind the input-gradient chain
        for i in range(n - 1, -1, -1):
            W, b = params[2 * i], params[2 * i + 1]
            if ctx.needs_input_grad[4 + 2 * i] or ctx.needs_input_grad[5 + 2 * i]:
                dW, db = ctx.direct[2 * i], ctx.direct[2 * i + 1]
                if dW is None or db is None:
                    dW, db = torch.zeros_like(W), torch.zeros_like(b)
                    grads[2 * i], grads[2 * i + 1] = dW, db
                dwg.dw(g, acts[i], dW, db)
            if i > 0:
                g = linear_bwd_input_raw(g, W, xact=acts[i])
            elif ctx.needs_input_grad[0]:
                if ctx.grad_cols is not None and ctx.grad_cols < W.shape[1]:
                    # the z-columns-only input gradient of the bottom layer is a sliver of a GEMM ([B, 256] x [256, Z]) and, like the
                    # weight gradients, needs nothing but the chain's last g: it rides in THEIR launch instead of paying its own
                    full = torch.empty(g.shape[0], W.shape[1], dtype=F32, device=g.device)
                    dwg.dx(g, W, xact=None, out=full[:, :ctx.grad_cols], cols=ctx.grad_cols)
                    g = full
                else:
                    g = linear_bwd_input_raw(g, W, xact=None)
            else:
                g = None
        dwg.launch()
        return (g, None, None, None) + tuple(grads)


def mlp_into(x, layers, out_buf, col0, grad_cols=None):
    if grad_cols is not None and x.requires_grad and type(x.grad_fn).__name__ != "_LatentPackedBackward":
        # the columns behind grad_cols of the returned input gradient are UNWRITTEN memory; only latent_packed (which reads the
        # first Z columns and nothing else) may be the producer of x
        raise RuntimeError("mlp_into(grad_cols=...): x must come straight from ops.latent_packed")
    flat = []
    for W, b in layers:
        flat += [W, b]
    return _MLPInto.apply(x, out_buf, int(col0), None if grad_cols is None else int(grad_cols), *flat)


def mlp(x, layers, last_linear):
    """layers: list of (weight [out,in], bias [out])."""
    flat = []
    for W, b in layers:
        flat += [W, b]
    return _MLP.apply(x, last_linear, *flat)


class _DenseScores(torch.autograd.Function):
    """p = rx @ E^T with E frozen (models/pivotcvae.py:274); used when the caller wants the dense logits."""

    @staticmethod
    def forward(ctx, rx, E):
        require_device(rx, E)
        ctx.save_for_backward(E)
        with mlp_arith(False):   # the dense logits a caller asks for are the exact fp32 ones, whatever the stacks compute in
            return linear_fwd_raw(rx, E, None, ACT_NONE)

    @staticmethod
    def backward(ctx, g):
        (E,) = ctx.saved_tensors
        with mlp_arith(False):
            return linear_bwd_input_raw(g, E), None


def dense_scores(rx, E):
    return _DenseScores.apply(rx, E)


# ------------------------------------------------------------------------------------------- K4
class _Reparam(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps, seed, offset):
        require_device(mu, logvar, eps)
        mu, logvar = _c2d(mu).contiguous(), _c2d(logvar).contiguous()
        B, Z = mu.shape
        z = torch.empty(B, Z, dtype=F32, device=mu.device)
        eps_used = torch.empty(B, Z, dtype=F32, device=mu.device)
        if eps is not None:
            eps = eps.to(F32).contiguous()
        check(lib().pcvae_reparam_fwd(ptr(mu, F32), ptr(logvar, F32), ptr(eps, F32) if eps is not None else None,
                                      seed, offset, ptr(z, F32), Z, ptr(eps_used, F32), B, Z, stream()),
              "reparam_fwd")
        ctx.save_for_backward(eps_used, logvar)
        ctx.mark_non_differentiable(eps_used)
        return z, eps_used

    @staticmethod
    def backward(ctx, gz, _geps):
        eps, logvar = ctx.saved_tensors
        gz = _c2d(gz)
        B, Z = eps.shape
        dmu = torch.zeros_like(eps)
        dlv = torch.zeros_like(eps)
        check(lib().pcvae_reparam_bwd(ptr(gz, F32), _ld(gz), ptr(eps, F32), ptr(logvar, F32), ptr(dmu, F32),
                                      ptr(dlv, F32), B, Z, stream()), "reparam_bwd")
        return dmu, dlv, None, None, None


def reparam(mu, logvar, eps=None, seed=0, offset=0):
    """-> (z, eps_used).  eps=None draws N(0,1) with the in-kernel Philox stream (seed, offset)."""
    return _Reparam.apply(mu, logvar, eps, int(seed), int(offset))


def philox_normal_(out, seed=0, offset=0):
    """Fill ``out`` with the N(0,1) stream reparam() would draw at (seed, offset)."""
    require_device(out)
    check(lib().pcvae_philox_normal(ptr(out, F32), out.numel(), int(seed), int(offset), stream()), "philox_normal")
    return out


# ------------------------------------------------------------------------------------------- K7
class _KLD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, lv, pmu, plv):
        require_device(mu, lv, pmu, plv)
        mu, lv, pmu, plv = (t.contiguous() for t in (mu, lv, pmu, plv))
        out = torch.empty((), dtype=F32, device=mu.device)
        check(lib().pcvae_kld_fwd(ptr(mu, F32), ptr(lv, F32), ptr(pmu, F32), ptr(plv, F32), mu.numel(), ptr(out, F32),
                                  stream()), "kld_fwd")
        ctx.save_for_backward(mu, lv, pmu, plv)
        return out

    @staticmethod
    def backward(ctx, g):
        mu, lv, pmu, plv = ctx.saved_tensors
        g = g.contiguous()
        outs = [torch.zeros_like(mu) if ctx.needs_input_grad[i] else None for i in range(4)]
        check(lib().pcvae_kld_bwd(ptr(mu, F32), ptr(lv, F32), ptr(pmu, F32), ptr(plv, F32), mu.numel(), ptr(g, F32), 1.0,
                                  *(ptr(o, F32) if o is not None else None for o in outs), stream()), "kld_bwd")
        return tuple(outs)


def kld(mu, logvar, pmu, plogvar):
    """-1/2 sum(1 + lv - plv - (exp(lv) + (mu - pmu)^2) / exp(plv)) as a 0-d tensor."""
    return _KLD.apply(mu, logvar, pmu, plogvar)


class _Latent(torch.autograd.Function):
    """reparam() and kld() of one posterior as ONE autograd node: the posterior's (mu, logvar) feed both, so separate nodes
    cost two backward kernels, six zero-fills and two gradient adds per step; here the backward is one kernel that writes
    all four gradients (pcvae_latent_bwd).  Forward values are those of reparam() / kld() (same kernels)."""

    @staticmethod
    def forward(ctx, mu, logvar, pmu, plogvar, eps, seed, offset):
        require_device(mu, logvar, pmu, plogvar)
        mu, logvar, pmu, plogvar = (_c2d(t).contiguous() for t in (mu, logvar, pmu, plogvar))
        B, Z = mu.shape
        z = torch.empty(B, Z, dtype=F32, device=mu.device)
        eps_used = torch.empty(B, Z, dtype=F32, device=mu.device)
        if eps is not None:
            eps = eps.to(F32).contiguous()
        check(lib().pcvae_reparam_fwd(ptr(mu, F32), ptr(logvar, F32), ptr(eps, F32) if eps is not None else None,
                                      seed, offset, ptr(z, F32), Z, ptr(eps_used, F32), B, Z, stream()), "reparam_fwd")
        k = torch.empty((), dtype=F32, device=mu.device)
        check(lib().pcvae_kld_fwd(ptr(mu, F32), ptr(logvar, F32), ptr(pmu, F32), ptr(plogvar, F32), mu.numel(), ptr(k, F32),
                                  stream()), "kld_fwd")
        ctx.save_for_backward(eps_used, mu, logvar, pmu, plogvar)
        ctx.mark_non_differentiable(eps_used)
        ctx.set_materialize_grads(False)   # an unused output arrives as None, not as a zero-filled tensor (a launch)
        return z, eps_used, k

    @staticmethod
    def backward(ctx, gz, _geps, gk):
        eps, mu, lv, pmu, plv = ctx.saved_tensors
        B, Z = eps.shape
        if gz is None:
            gz = torch.zeros_like(eps)
        gz = _c2d(gz)
        outs = [torch.empty_like(eps) for _ in range(4)]
        gk_ptr, gk_host = (ptr(gk.contiguous(), F32), 1.0) if gk is not None else (None, 0.0)
        check(lib().pcvae_latent_bwd(ptr(gz, F32), _ld(gz), ptr(eps, F32), ptr(mu, F32), ptr(lv, F32), ptr(pmu, F32),
                                     ptr(plv, F32), gk_ptr, gk_host, *(ptr(o, F32) for o in outs), B, Z, stream()),
              "latent_bwd")
        return outs[0], outs[1], outs[2], outs[3], None, None, None


def latent(mu, logvar, pmu, plogvar, eps=None, seed=0, offset=0):
    """-> (z, eps_used, kld): reparam(mu, logvar, eps, seed, offset) and kld(mu, logvar, pmu, plogvar) with a fused backward."""
    return _Latent.apply(mu, logvar, pmu, plogvar, eps, int(seed), int(offset))


# -------------------------------------------------------------------------------------- K5 / K6
CATALOG_DIMS = (16, 32, 64, 128, 256)   # widths the catalog kernels are instantiated for


def _padded_width(D):
    """narrowest supported catalog width >= D (the reference's default --dim is 8, train_generative.py:302)"""
    for w in CATALOG_DIMS:
        if w >= D:
            return w
    raise ValueError(f"catalog kernels support D <= {CATALOG_DIMS[-1]}, got D={D}")


class CatalogTable:
    """The frozen item table E[N, D] plus, lazily and cached per table version: its bf16 copy (bf16 MFMA mode), the
    interleaved [N, 2D] bf16 hi | lo image (bf16x3 mode), and a zero-padded fp32 copy for widths the kernels are not
    instantiated for (zero columns change neither a logit nor the gradient of the real columns)."""

    def __init__(self, weight):
        self.weight = weight
        self._ver = None
        self._hi = self._x3 = self._x6 = None
        self._pad = {}
        self._emax = 0.0

    def _refresh(self):
        w = self.weight
        key = (w.data_ptr(), w._version, tuple(w.shape))
        if self._ver != key:
            self._hi = self._x3 = self._x6 = None
            self._pad = {}
            self._emax = 0.0
            self._ver = key

    def padded(self, width=None):
        """-> (fp32 table of a supported width, that width); ``width``: pad to exactly this many columns (the bf16x3 kernel's 128)"""
        self._refresh()
        w = self.weight
        D = w.shape[1]
        Dp = _padded_width(D) if width is None else int(width)
        if Dp == D:
            return w, D
        if Dp not in self._pad:
            pad = torch.zeros(w.shape[0], Dp, dtype=F32, device=w.device)
            copy2d(w.detach(), pad[:, :D])
            self._pad[Dp] = pad
        return self._pad[Dp], Dp

    def _max_norm(self, w32):
        if self._emax == 0.0:
            # max row norm: lets the bf16 kernels skip the running max where |logit| provably stays small.
            # One-off per table version (frozen table), like the bf16 copies themselves.
            self._emax = float(w32.pow(2).sum(1).max().sqrt().item()) * (1.0 + 1e-6)

    def operands(self, prec):
        """-> (E, E_lo) as pcvae_catalog_ce takes them for this precision mode"""
        self._refresh()
        w = self.weight
        if prec == PREC_F32:
            return self.padded()[0], None
        w32 = w.detach().contiguous()
        self._max_norm(w32)
        if prec == PREC_BF16X3:
            w32 = self.padded(x3_width(w.shape[1]))[0].detach()   # narrower tables ride the 128-wide kernel on zero columns
            if self._x3 is None:
                x3 = torch.empty(w32.shape[0], 2 * w32.shape[1], dtype=torch.int16, device=w.device)
                check(lib().pcvae_split_bf16x2(ptr(w32, F32), w32.shape[0], w32.shape[1], ptr(x3), stream()), "split_bf16x2")
                self._x3 = x3
            return self._x3, w32   # E_lo carries the exact fp32 table: row blocks with large norms run the f32 kernel
        if prec == PREC_BF16X6:
            w32 = self.padded(x6_width(w.shape[1]))[0].detach()   # narrower tables ride the 128-wide kernel on zero columns
            if self._x6 is None:
                x6 = torch.empty(w32.shape[0], 3 * w32.shape[1], dtype=torch.int16, device=w.device)
                check(lib().pcvae_split_bf16x3(ptr(w32, F32), w32.shape[0], w32.shape[1], ptr(x6), stream()), "split_bf16x3")
                self._x6 = x6
            return self._x6, w32
        if self._hi is None:
            hi = torch.empty(w32.shape, dtype=torch.int16, device=w.device)
            check(lib().pcvae_split_bf16(ptr(w32, F32), w32.numel(), ptr(hi), None, stream()), "split_bf16")
            self._hi = hi
        return self._hi, None

    def e_max_norm(self, prec):
        return 0.0 if prec == PREC_F32 else self._emax


_ws_cache = {}
# bench.py installs (begin, end) callables here to bracket the dominant kernel with HIP events
CATALOG_CE_TIMING = None
# ... and here around the pivot-selection kernels (catalog_argmax / catalog_sample: the pt / spt / sgt / spi rules)
PIVOT_TIMING = None


_ws_holder = None   # set by workspace_holder(): a dict that owns the scratch buffers instead of the module cache


class workspace_holder:
    """``with workspace_holder(d):`` every catalog call inside takes its scratch buffer from the dict ``d`` (grow-only, keyed like
    the module cache).  A hipGraph bakes the scratch pointer into its kernels: the Trainer captures under its OWN holder and keeps
    it alive with the graph, so no later, larger call on the device can free memory a replay still writes to."""

    def __init__(self, holder):
        self.holder = holder

    def __enter__(self):
        global _ws_holder
        self.prev, _ws_holder = _ws_holder, self.holder
        return self.holder

    def __exit__(self, *exc):
        global _ws_holder
        _ws_holder = self.prev


def _workspace(device, nbytes, tag="catalog", zero=False):
    """Grow-only scratch buffer per (device, stream, user): the C ABI never allocates, and two streams never share scratch.  The
    catalog kernels and the grouped GEMMs keep separate buffers (`tag`): the latter holds counters that must be ZERO between
    launches (every launch leaves them so; `zero` clears a new allocation)."""
    cache = _ws_cache if _ws_holder is None else _ws_holder
    key = (device, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = (torch.zeros if zero else torch.empty)(int(nbytes), dtype=torch.uint8, device=device)
        cache[key] = buf
    return buf


def _as_table(E):
    return E if isinstance(E, CatalogTable) else CatalogTable(E)


BF16_DIMS = (64, 128, 256)  # the bf16 MFMA kernels exist for these widths; narrower tables are tiny: exact f32 path
X3_DIMS = (128, 256)        # widths the bf16x3 (fp32-equivalent) kernel is instantiated for (256: two images of 128 dims) ...
X3_MAX_PADDED = 128         # ... and any narrower table runs it on zero columns (a zero column adds exactly 0 to every product of the
#                             three-MFMA split: same logits, same gradient in the real columns).  D = 64 does 2x, D = 32 4x the
#                             necessary MFMAs and is still 2.2x / 1.6x faster than the exact f32-MFMA kernel at its tolerances


X6_DIMS = (128,)            # bf16x6 (three bf16 components per operand = the fp32 operand exactly, 6 MFMAs per product): D = 128;
#                             narrower tables ride it on zero columns; D = 256 has no such kernel (its ring would not fit a CU's LDS)


def x6_width(D):
    """table width the bf16x6 kernel runs a D-wide catalog at (None: no bf16x6 route for this width: exact f32 then)"""
    return 128 if D <= 128 else None


def split_width(prec, D):
    """x3_width / x6_width by precision mode"""
    return x6_width(D) if prec == PREC_BF16X6 else x3_width(D)


def x3_width(D):
    """table width the bf16x3 kernel runs a D-wide catalog at (None: no bf16x3 route for this width)"""
    if D in X3_DIMS:
        return D
    return X3_MAX_PADDED if D < X3_MAX_PADDED else None


def _word(v):
    """a seed / stream position given by value (int) or as a DEVICE word (a 1-element int64 tensor: a hipGraph-replayed step reads
    it at run time, kernel arguments are frozen at capture) -> (by-value argument, device pointer or None)"""
    if torch.is_tensor(v):
        if v.numel() != 1 or v.dtype not in (torch.int64, torch.uint64) or not v.is_cuda:
            raise TypeError("a device-word seed is a 1-element int64 tensor on the ROCm device")
        return 0, ctypes.c_void_p(v.data_ptr())
    return int(v), None


def set_words_(words, a, b):
    """words[0] = a, words[1] = b on the current stream (one tiny launch; the step-dependent words of a replayed hipGraph)"""
    check(lib().pcvae_set_words(ptr(words, torch.int64), int(a) & 0xFFFFFFFFFFFFFFFF, int(b) & 0xFFFFFFFFFFFFFFFF, stream()), "set_words")
    return words


SPARSE_MAX_KEEP_PROB = 0.03   # above this the dense masked kernel (one pass over the catalog) is the cheaper one


def sparse_ce_applies(keep_prob, N):
    """does catalog_ce take the sparse (kept rows only) path for this keep probability?  (train_generative.py:44: n_neg = 1000)"""
    return 0.0 < keep_prob <= SPARSE_MAX_KEEP_PROB and N < 2 ** 31 - 1


def gather_rows_are_bf16(model):
    """do this model's gather kernels read bf16 table rows?  (an explicit switch - BaseCVAE.set_gather_rows("bf16") - never implied
    by the catalog contraction's arithmetic: the validation loss of the epoch loop and get_gen_loss's candidate branch stay the
    reference's fp32 arithmetic unless asked otherwise)"""
    return bool(getattr(model, "gather_rows_bf16", False)) and model.docEmbed.weight.shape[1] in BF16_DIMS


def _gather_table(table, D0, prec):
    """the table the gather kernels (sparse K5, K9) read -> (tensor, width, PREC_F32 | PREC_BF16).  ``prec`` = PREC_BF16 (the caller
    asked for bf16 rows EXPLICITLY: half the gathered bytes; widened exactly, fp32 products and sums) and a width that has a bf16
    table; the fp32 table - the reference's arithmetic - otherwise"""
    if prec == PREC_BF16 and D0 in BF16_DIMS:
        return table.operands(PREC_BF16)[0], D0, PREC_BF16
    E, D = table.padded()
    return E, D, PREC_F32


def catalog_ce_sparse_raw(rx, table, target, keep_prob, seed=0, row_offset=0, want_dx=True, dx_scale=1.0, prec=PREC_F32):
    """catalog_ce_raw for keep_prob << 1: only the kept rows of the table are read (pcvae_catalog_ce_sparse).
    -> (nll [R], lse [R], dx [R, D] or None).  Exact fp32 of the fp32 table, except under ``prec`` = bf16 (_gather_table)."""
    table = _as_table(table)
    require_device(rx, table.weight, target)
    rx = _c2d(rx).contiguous()
    R, D0 = rx.shape
    E, D, gprec = _gather_table(table, D0, prec)
    rx = _pad_cols(rx, D)
    N = E.shape[0]
    target = target.reshape(-1).to(torch.int64).contiguous()
    if target.numel() != R:
        raise ValueError("catalog_ce: one target per row expected")
    nll = torch.empty(R, dtype=F32, device=rx.device)
    lse = torch.empty(R, dtype=F32, device=rx.device)
    dx = torch.empty(R, D, dtype=F32, device=rx.device) if want_dx else None
    timing = CATALOG_CE_TIMING
    tok = timing[0]() if timing else None
    seed_val, seed_dev = _word(seed)
    check(lib().pcvae_catalog_ce_sparse_scaled(ptr(rx, F32), R, ptr(E), gprec, N, D, ptr(target), float(keep_prob), seed_val,
                                               int(row_offset), ptr(nll, F32), ptr(lse, F32), ptr(dx), float(dx_scale), seed_dev,
                                               stream()), "catalog_ce_sparse")
    if timing:
        timing[1](tok)
    if dx is not None and D != D0:
        dx = dx[:, :D0]
    return nll, lse, dx


def effective_precision(prec, D):
    """bf16 kernels exist for D in BF16_DIMS, the bf16x3 kernel for D <= 128 (natively at 128, narrower tables zero-padded);
    everything else computes in exact f32"""
    if prec == PREC_BF16 and D in BF16_DIMS:
        return PREC_BF16
    if prec == PREC_BF16X3 and x3_width(D) is not None:
        return PREC_BF16X3
    if prec == PREC_BF16X6 and x6_width(D) is not None:
        return PREC_BF16X6
    return PREC_F32


def _pad_cols(x, Dp):
    """[R, D] -> [R, Dp] with zero columns behind (a no-op when D == Dp)"""
    R, D = x.shape
    if D == Dp:
        return x
    out = torch.zeros(R, Dp, dtype=F32, device=x.device)
    copy2d(x, out[:, :D])
    return out


def catalog_ce_raw(rx, table, target, keep_prob=1.0, seed=0, row_offset=0, keep_mask=None, prec=PREC_F32,
                   want_dx=True, dx_scale=1.0, gather_bf16=False):
    """-> (nll [R], lse [R], dx [R, D] * dx_scale or None); see pcvae_catalog_ce in include/pcvae.h.  ``gather_bf16``: the sparse
    kept-rows kernel (keep_prob << 1) reads bf16 table rows (an explicit switch; default: the fp32 table whatever ``prec`` is)."""
    table = _as_table(table)
    require_device(rx, table.weight, target, keep_mask)
    rx = _c2d(rx).contiguous()
    R, D0 = rx.shape
    N = table.weight.shape[0]
    target = target.reshape(-1).to(torch.int64).contiguous()
    if target.numel() != R:
        raise ValueError("catalog_ce: one target per row expected")
    if keep_mask is not None:
        keep_mask = keep_mask.to(torch.uint8).contiguous()
        if tuple(keep_mask.shape) != (R, N):
            raise ValueError("catalog_ce: keep_mask must be [R, N]")
    prec = effective_precision(prec, D0)
    if keep_mask is None and keep_prob < 1.0 and sparse_ce_applies(keep_prob, N):
        return catalog_ce_sparse_raw(rx, table, target, keep_prob, seed, row_offset, want_dx, dx_scale,
                                     PREC_BF16 if gather_bf16 else PREC_F32)
    if torch.is_tensor(seed):
        if keep_mask is None and keep_prob < 1.0:
            raise RuntimeError("a device-word mask seed (hipGraph replay) exists for the sparse kept-rows kernel only "
                               f"(keep_prob <= {SPARSE_MAX_KEEP_PROB})")
        seed = 0   # no in-kernel draw in this call
    if keep_mask is not None or keep_prob < 1.0:
        prec = PREC_F32 if prec in (PREC_BF16X3, PREC_BF16X6) else prec   # masked calls: the split-bf16 kernels are max-free / mask-free
    E, E_lo = table.operands(prec)
    D = split_width(prec, D0) if prec in (PREC_BF16X3, PREC_BF16X6) else _padded_width(D0)
    rx = _pad_cols(rx, D)
    nll = torch.empty(R, dtype=F32, device=rx.device)
    lse = torch.empty(R, dtype=F32, device=rx.device)
    dx = torch.empty(R, D, dtype=F32, device=rx.device) if want_dx else None
    nbytes = lib().pcvae_catalog_ws_bytes(R, N, D, 1 if want_dx else 0)
    ws = _workspace(rx.device, nbytes)
    timing = CATALOG_CE_TIMING
    tok = timing[0]() if timing else None
    check(lib().pcvae_catalog_ce_scaled(ptr(rx, F32), R, ptr(E), ptr(E_lo), N, D, prec, table.e_max_norm(prec), ptr(target),
                                        float(keep_prob), int(seed), int(row_offset), ptr(keep_mask), ptr(nll, F32), ptr(lse, F32),
                                        ptr(dx), float(dx_scale), ptr(ws), ws.numel(), stream()), "catalog_ce")
    if timing:
        timing[1](tok)
    if dx is not None and D != D0:
        dx = dx[:, :D0]
    return nll, lse, dx


_unit_seeds = {}   # data_ptr -> weak reference to the seed tensor (a dead tensor's address may be reused: then it no longer counts)


def register_unit_seed(t):
    """``t``: a 0-d device tensor holding exactly 1.0 that the caller will pass as the upstream gradient of a ``unit_upstream`` loss
    (Trainer does).  Only then does backward hand the saved direction on without a scaling launch; any other upstream gradient
    (a scaled loss, accumulation with loss / k) is applied."""
    import weakref
    for k in [k for k, r in _unit_seeds.items() if r() is None]:
        del _unit_seeds[k]
    _unit_seeds[t.data_ptr()] = weakref.ref(t)
    return t


def _is_unit_seed(g):
    r = _unit_seeds.get(g.data_ptr())
    t = r() if r is not None else None
    return t is not None and t.data_ptr() == g.data_ptr()


class _CatalogCE(torch.autograd.Function):
    """mean-reduced (times ``inv_count``) full-catalog softmax CE; backward = saved direction * upstream.
    ``unit_upstream``: the caller intends to seed the backward of this loss with a constant 1 registered through
    ``register_unit_seed`` (Trainer does: it seeds (rec, KLD) with (1, beta)); the kernel then writes the direction times inv_count
    and backward hands it on without a scaling launch.  Any other upstream gradient is still applied correctly (one launch)."""

    @staticmethod
    def forward(ctx, rx, table, target, keep_prob, seed, row_offset, keep_mask, prec, inv_count, unit_upstream, gather_bf16):
        want_dx = rx.requires_grad
        nll, _lse, dx = catalog_ce_raw(rx.detach(), table, target, keep_prob, seed, row_offset, keep_mask, prec, want_dx,
                                       dx_scale=float(inv_count) if unit_upstream else 1.0, gather_bf16=gather_bf16)
        out = torch.empty((), dtype=F32, device=rx.device)
        check(lib().pcvae_sum(ptr(nll, F32), nll.numel(), float(inv_count), ptr(out, F32), stream()), "sum")
        ctx.inv_count = float(inv_count)
        ctx.unit = bool(unit_upstream)
        if want_dx:
            ctx.save_for_backward(dx)
        return out

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        if ctx.unit and _is_unit_seed(g):   # the registered constant 1: the kernel already wrote dx * inv_count
            return (dx,) + (None,) * 10
        g = g.contiguous()
        out = torch.empty_like(dx)
        # unit_upstream but some OTHER upstream gradient: dx is pre-scaled by inv_count, only g is left to apply
        check(lib().pcvae_scale_rows(ptr(dx, F32), _ld(dx), ptr(out, F32), _ld(out), dx.shape[0], dx.shape[1],
                                     ptr(g, F32), 1.0 if ctx.unit else ctx.inv_count, stream()), "scale_rows")
        return (out,) + (None,) * 10


def catalog_ce(rx, table, target, keep_prob=1.0, seed=0, row_offset=0, keep_mask=None, prec=PREC_F32, inv_count=None,
               unit_upstream=False, gather_bf16=False):
    """CrossEntropyLoss(downsample(rx @ E^T), target) without the [R, N] logits (train_generative.py:59).

    ``inv_count`` defaults to 1/R (the 'mean'); data-parallel ranks pass 1/(R_local * world_size).
    """
    R = rx.shape[0]
    return _CatalogCE.apply(rx, _as_table(table), target, keep_prob, seed, row_offset, keep_mask, prec,
                            (1.0 / R) if inv_count is None else inv_count, unit_upstream, gather_bf16)


# exact argmax through bf16 screening pays off once the catalog is large; below this the plain f32 kernel is used
SCREENED_MIN_ITEMS = 32768


def catalog_argmax(x, table, prec=PREC_F32, return_best=False, screened=None):
    """first-index argmax_n <x_r, E_n> (models/cvae.py:97-101) -> int64 [R].

    Always the exact fp32 answer (bit-exact ids against the fp32 oracle), independent of ``prec`` (kept for call
    compatibility with the loss ops).  With D in (64, 128, 256) and a large catalog the same exact result is produced several
    times faster by bf16 screening + fp32 rescoring of the candidates (PCVAE_PREC_SCREENED); ``screened`` forces
    (True) or forbids (False) that route."""
    table = _as_table(table)
    require_device(x, table.weight)
    x = _c2d(x.detach()).contiguous()
    R, D = x.shape
    N = table.weight.shape[0]
    # ids are index work: whatever precision the training loss runs in, the argmax is the exact fp32 one
    use = (D in BF16_DIMS and N >= SCREENED_MIN_ITEMS) if screened is None else bool(screened)
    if use and D not in BF16_DIMS:
        raise ValueError(f"screened argmax exists for D in {BF16_DIMS} only")
    if use:
        E, _ = table.operands(PREC_BF16)
        E_lo = table.weight
        mode, emax = PREC_SCREENED, table.e_max_norm(PREC_BF16)
    else:
        E, D = table.padded()   # zero columns: fmaf(0, 0, acc) == acc, ids and scores stay bit-exact
        x = _pad_cols(x, D)
        E_lo, mode, emax = None, PREC_F32, 0.0
    idx = torch.empty(R, dtype=torch.int64, device=x.device)
    best = torch.empty(R, dtype=F32, device=x.device) if return_best else None
    nbytes = lib().pcvae_catalog_ws_bytes(R, N, D, 0)
    ws = _workspace(x.device, nbytes)
    timing = PIVOT_TIMING
    tok = timing[0]() if timing else None
    check(lib().pcvae_catalog_argmax(ptr(x, F32), R, ptr(E), ptr(E_lo), N, D, mode, float(emax), ptr(idx), ptr(best), ptr(ws),
                                     ws.numel(), stream()), "catalog_argmax")
    if timing:
        timing[1](tok)
    return (idx, best) if return_best else idx


def catalog_sample(x, table, seed=0, row_offset=0):
    """idx[r] ~ Categorical(sigmoid(<x_r, E_n>)) (models/pivotcvae.py:349-351), drawn by REJECTION sampling: propose a uniform item,
    accept it with probability sigmoid(score) - exactly the reference's distribution for ~2 dot products per row instead of the
    [R, N] score matrix (csrc/catalog_sample.hip; rows that reject 512 proposals in a row fall back to the exact Gumbel-max
    kernel).  Scores are exact fp32 of the fp32 table whatever arithmetic the training loss runs in (there is no precision
    argument: round 5's ignored ``prec`` is gone)."""
    table = _as_table(table)
    require_device(x, table.weight)
    x = _c2d(x.detach()).contiguous()
    R = x.shape[0]
    N = table.weight.shape[0]
    E, D = table.padded()   # sampling scores are fp32 whatever the loss precision is
    x = _pad_cols(x, D)
    E_lo = None
    idx = torch.empty(R, dtype=torch.int64, device=x.device)
    ws = _workspace(x.device, lib().pcvae_catalog_ws_bytes(R, N, D, 0))
    timing = PIVOT_TIMING
    tok = timing[0]() if timing else None
    if isinstance(row_offset, (tuple, list)):    # (by-value part, device word added to it): a hipGraph-replayed step
        off_val, off_dev = int(row_offset[0]), _word(row_offset[1])[1]
    else:
        off_val, off_dev = int(row_offset), None
    check(lib().pcvae_catalog_sample_at(ptr(x, F32), R, ptr(E), ptr(E_lo), N, D, PREC_F32, int(seed), off_val, off_dev,
                                        ptr(idx), ptr(ws), ws.numel(), stream()), "catalog_sample")
    if timing:
        timing[1](tok)
    return idx


# ------------------------------------------------------------------------------------------- K9
class _CandidateScores(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rx, E, cand):
        require_device(rx, E, cand)
        rx = _c2d(rx).contiguous()
        R, D = rx.shape
        cand = cand.reshape(R, -1).to(torch.int64).contiguous()
        Cn = cand.shape[1]
        p = torch.empty(R, Cn, dtype=F32, device=rx.device)
        check(lib().pcvae_candidate_scores(ptr(rx, F32), R, ptr(E, F32), E.shape[0], D, ptr(cand), Cn, ptr(p, F32),
                                           stream()), "candidate_scores")
        ctx.save_for_backward(E, cand)
        return p

    @staticmethod
    def backward(ctx, g):
        E, cand = ctx.saved_tensors
        g = _c2d(g).contiguous()
        R, Cn = g.shape
        D = E.shape[1]
        drx = torch.empty(R, D, dtype=F32, device=g.device)
        check(lib().pcvae_candidate_scores_bwd(ptr(g, F32), R, ptr(E, F32), E.shape[0], D, ptr(cand), Cn,
                                               ptr(drx, F32), stream()), "candidate_scores_bwd")
        return drx, None, None


def candidate_scores(rx, E, cand):
    """p[r, c] = <E[cand[r, c]], rx_r> (models/pivotcvae.py:265-271)."""
    return _CandidateScores.apply(rx, E, cand)


def candidate_draw(slates, n_items, n_candidate, seed=0, row_offset=0, raw=None):
    """Candidate sets of the sampled-softmax path on the device (data_loader.py:46-58).

    slates [B, S] int64 -> (sample_candidates [B, S, Cn] int64, sample_targets [B, S] int64): Cn uniform ids per slot; the
    slot's true item is the target - the first column that holds it, or column 0 overwritten with it.  ``raw`` [B, S, Cn]
    replays a recorded draw (parity with the reference's numpy stream, which cannot be matched on the device)."""
    require_device(slates, raw)
    B, S = slates.shape
    f = slates.reshape(-1).to(torch.int64).contiguous()
    Cn = int(n_candidate)
    cand = torch.empty(B * S, Cn, dtype=torch.int64, device=slates.device)
    tgt = torch.empty(B * S, dtype=torch.int64, device=slates.device)
    if raw is not None:
        raw = raw.reshape(B * S, Cn).to(torch.int64).contiguous()
    check(lib().pcvae_candidate_draw(ptr(f), B * S, int(n_items), Cn, int(seed), int(row_offset), ptr(raw), ptr(cand), ptr(tgt),
                                     stream()), "candidate_draw")
    return cand.view(B, S, Cn), tgt.view(B, S)


def candidate_ce_raw(rx, table, n_candidate=None, feature=None, seed=0, row_offset=0, cand=None, cand_target=None, want_dx=True,
                     dx_scale=1.0, want_target=False, prec=PREC_F32, n_items=None):
    """The candidate-set softmax CE in ONE launch (pcvae_candidate_ce): -> (nll [R], lse [R], dx [R, D] * dx_scale or None,
    target column [R] or None).  Either ``cand`` [R, Cn] + ``cand_target`` [R] (sets as given: a batch of the reference's dataset,
    a recorded draw) or ``feature`` [R] + ``n_candidate`` (sets drawn in-kernel from the stream of ``candidate_draw(slates,
    n_items, ...)``: uniform ids in [0, n_items), the dataset's ``max_iid + 1`` - data_loader.py:23, :46; None = the table's rows)."""
    table = _as_table(table)
    require_device(rx, table.weight, feature, cand, cand_target)
    rx = _c2d(rx).contiguous()
    R, D0 = rx.shape
    E, D, gprec = _gather_table(table, D0, prec)
    rx = _pad_cols(rx, D)
    N = E.shape[0]
    if cand is not None:
        cand = cand.reshape(R, -1).to(torch.int64).contiguous()
        Cn = cand.shape[1]
        if cand_target is None:
            raise ValueError("candidate_ce: given candidate sets need their target columns (sample_targets)")
        cand_target = cand_target.reshape(-1).to(torch.int64).contiguous()
        if cand_target.numel() != R:
            raise ValueError("candidate_ce: one target column per row expected")
        feature = None
    else:
        if feature is None or n_candidate is None:
            raise ValueError("candidate_ce: pass the slots' true items and n_candidate, or candidate sets")
        Cn = int(n_candidate)
        feature = feature.reshape(-1).to(torch.int64).contiguous()
        if feature.numel() != R:
            raise ValueError("candidate_ce: one true item per row expected")
        if n_items is not None and not 0 < int(n_items) <= N:
            raise ValueError(f"candidate_ce: n_items={n_items} must lie in (0, {N}] (the table's row count)")
    nll = torch.empty(R, dtype=F32, device=rx.device)
    lse = torch.empty(R, dtype=F32, device=rx.device)
    dx = torch.empty(R, D, dtype=F32, device=rx.device) if want_dx else None
    tcol = torch.empty(R, dtype=torch.int64, device=rx.device) if want_target else None
    timing = CATALOG_CE_TIMING   # the step's reconstruction kernel, whichever it is (bench.py times it on the launch stream)
    tok = timing[0]() if timing else None
    seed_val, seed_dev = _word(seed)
    check(lib().pcvae_candidate_ce(ptr(rx, F32), R, ptr(E), gprec, N, D, Cn, ptr(feature), seed_val, int(row_offset), ptr(cand),
                                   ptr(cand_target), ptr(nll, F32), ptr(lse, F32), ptr(dx), float(dx_scale), ptr(tcol), seed_dev,
                                   0 if n_items is None else int(n_items), stream()), "candidate_ce")
    if timing:
        timing[1](tok)
    if dx is not None and D != D0:
        dx = dx[:, :D0]
    return nll, lse, dx, tcol


class _CandidateCE(torch.autograd.Function):
    """mean-reduced (times ``inv_count``) candidate-set softmax CE; backward = saved direction * upstream (``unit_upstream`` as in
    _CatalogCE: the kernel writes the direction times inv_count and a registered constant-1 seed hands it on without a launch)."""

    @staticmethod
    def forward(ctx, rx, table, n_candidate, feature, seed, row_offset, cand, cand_target, inv_count, unit_upstream, prec, n_items):
        want_dx = rx.requires_grad
        nll, _lse, dx, _t = candidate_ce_raw(rx.detach(), table, n_candidate, feature, seed, row_offset, cand, cand_target, want_dx,
                                             dx_scale=float(inv_count) if unit_upstream else 1.0, prec=prec, n_items=n_items)
        out = torch.empty((), dtype=F32, device=rx.device)
        check(lib().pcvae_sum(ptr(nll, F32), nll.numel(), float(inv_count), ptr(out, F32), stream()), "sum")
        ctx.inv_count = float(inv_count)
        ctx.unit = bool(unit_upstream)
        if want_dx:
            ctx.save_for_backward(dx)
        return out

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        if ctx.unit and _is_unit_seed(g):
            return (dx,) + (None,) * 11
        g = g.contiguous()
        out = torch.empty_like(dx)
        check(lib().pcvae_scale_rows(ptr(dx, F32), _ld(dx), ptr(out, F32), _ld(out), dx.shape[0], dx.shape[1],
                                     ptr(g, F32), 1.0 if ctx.unit else ctx.inv_count, stream()), "scale_rows")
        return (out,) + (None,) * 11


def candidate_ce(rx, table, n_candidate=None, feature=None, seed=0, row_offset=0, cand=None, cand_target=None, inv_count=None,
                 unit_upstream=False, prec=PREC_F32, n_items=None):
    """CrossEntropyLoss(bmm(docEmbed(candidates), rx), sample_targets) (models/pivotcvae.py:265-271, train_generative.py:52-57)
    without the [R, Cn] ids, the [R, Cn, D] rows or the [R, Cn] logits: loss and d rx from one launch.

    ``inv_count`` defaults to 1/R (the 'mean'); data-parallel ranks pass 1/(R_local * world_size)."""
    R = rx.shape[0]
    return _CandidateCE.apply(rx, _as_table(table), n_candidate, feature, seed, row_offset, cand, cand_target,
                              (1.0 / R) if inv_count is None else inv_count, unit_upstream, prec, n_items)


def urm_forward(E, item_bias, U, user_bias, slates, users, pos_bias=None, pos_dep=None, mr_factor=None):
    """URM / URM_P / URM_P_MR.core_forward (env/response_model.py:129-150, 286-295, 315-323) -> [B, S] scores."""
    require_device(E, item_bias, U, user_bias, slates, users, pos_bias, pos_dep)
    B, S = slates.shape
    D = E.shape[1]
    sl = slates.to(torch.int64).contiguous()
    us = users.reshape(-1).to(torch.int64).contiguous()
    if us.numel() != B:
        raise ValueError("urm_forward: one user per slate expected")
    out = torch.empty(B, S, dtype=F32, device=E.device)
    check(lib().pcvae_urm_forward(ptr(E.contiguous(), F32), ptr(item_bias.reshape(-1).contiguous(), F32), E.shape[0],
                                  ptr(U.contiguous(), F32), ptr(user_bias.reshape(-1).contiguous(), F32), U.shape[0], ptr(sl), ptr(us),
                                  ptr(pos_bias.contiguous(), F32) if pos_bias is not None else None,
                                  ptr(pos_dep.contiguous(), F32) if pos_dep is not None else None,
                                  float(mr_factor or 0.0), 1 if mr_factor is not None else 0, B, S, D, ptr(out, F32), stream()),
          "urm_forward")
    return out


class _DownsampleDense(torch.autograd.Function):
    """pred * mask with mask = onehot(target) OR Bernoulli(keep_prob) (train_generative.py:36-42), the mask drawn in the kernel."""

    @staticmethod
    def forward(ctx, pred, slate, keep_prob, seed, row_offset):
        require_device(pred, slate)
        pred = _c2d(pred)
        R, N = pred.shape
        slate = slate.reshape(-1).to(torch.int64).contiguous()
        out = torch.empty(R, N, dtype=F32, device=pred.device)
        check(lib().pcvae_downsample_dense(ptr(pred, F32), _ld(pred), ptr(slate), R, N, float(keep_prob), int(seed), int(row_offset),
                                           ptr(out, F32), N, stream()), "downsample_dense")
        ctx.save_for_backward(slate)
        ctx.args = (float(keep_prob), int(seed), int(row_offset))
        return out

    @staticmethod
    def backward(ctx, g):   # the same mask applied to the upstream gradient
        (slate,) = ctx.saved_tensors
        g = _c2d(g)
        R, N = g.shape
        out = torch.empty(R, N, dtype=F32, device=g.device)
        check(lib().pcvae_downsample_dense(ptr(g, F32), _ld(g), ptr(slate), R, N, ctx.args[0], ctx.args[1], ctx.args[2], ptr(out, F32), N,
                                           stream()), "downsample_dense")
        return out, None, None, None, None


def downsample_dense(pred, slate, keep_prob, seed=0, row_offset=0):
    return _DownsampleDense.apply(pred, slate, keep_prob, seed, row_offset)


# ------------------------------------------------------- in-loop evaluation (response model)
def normalize_rows_(x):
    """x[r, :] /= max(||x[r, :]||, 1e-12) in place (F.normalize(p=2, dim=1))."""
    require_device(x)
    check(lib().pcvae_normalize_rows(ptr(x, F32), _ld(x), x.shape[0], x.shape[1], stream()), "normalize_rows")
    return x


def click_stats(logits):
    """-> (nc [B] = sum_s sigmoid(logits), stats [3] = (min, mean, max) of nc)   (train_generative.py:185-190)."""
    require_device(logits)
    logits = _c2d(logits).contiguous()
    B, S = logits.shape
    nc = torch.empty(B, dtype=F32, device=logits.device)
    out = torch.empty(3, dtype=F32, device=logits.device)
    check(lib().pcvae_click_stats(ptr(logits, F32), B, S, ptr(nc, F32), ptr(out, F32), stream()), "click_stats")
    return nc, out


def philox_randint(n, hi, device, seed=0, offset=0):
    """n uniform integers in [0, hi) from the Philox stream (seed, offset) -> int64 [n]."""
    out = torch.empty(int(n), dtype=torch.int64, device=device)
    require_device(out)
    check(lib().pcvae_philox_randint(ptr(out), out.numel(), int(hi), int(seed), int(offset), stream()), "philox_randint")
    return out


class _DenseCE(torch.autograd.Function):
    """mean softmax cross-entropy over a small dense class axis (the candidate path), fused loss + gradient."""

    @staticmethod
    def forward(ctx, p, target, inv_count):
        require_device(p, target)
        p = _c2d(p)
        R, C = p.shape
        target = target.reshape(-1).to(torch.int64).contiguous()
        nll = torch.empty(R, dtype=F32, device=p.device)
        dp = torch.empty(R, C, dtype=F32, device=p.device) if p.requires_grad else None
        check(lib().pcvae_dense_ce(ptr(p, F32), _ld(p), R, C, ptr(target), ptr(nll, F32), ptr(dp), C, stream()), "dense_ce")
        out = torch.empty((), dtype=F32, device=p.device)
        check(lib().pcvae_sum(ptr(nll, F32), R, float(inv_count), ptr(out, F32), stream()), "sum")
        ctx.inv = float(inv_count)
        if dp is not None:
            ctx.save_for_backward(dp)
        return out

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(dp)
        check(lib().pcvae_scale_rows(ptr(dp, F32), _ld(dp), ptr(out, F32), _ld(out), dp.shape[0], dp.shape[1], ptr(g, F32),
                                     ctx.inv, stream()), "scale_rows")
        return out, None, None


def dense_ce(p, target, inv_count=None):
    """nn.CrossEntropyLoss()(p, target) for a small dense [R, C] logits tensor.  ``inv_count`` replaces the 1/R of the mean
    (a data-parallel rank passes 1/(R_local * world_size))."""
    return _DenseCE.apply(p, target, (1.0 / p.shape[0]) if inv_count is None else inv_count)


def elbo_pack_(rec, kld, beta, out):
    """out[0..2] = (rec + beta * kld, rec, kld) - the logged terms of a step as one record (pcvae_elbo_pack)"""
    require_device(rec, kld, out)
    check(lib().pcvae_elbo_pack(ptr(rec, F32), ptr(kld, F32), float(beta), ptr(out, F32), stream()), "elbo_pack")
    return out


def zero_(t):
    """t[...] = 0 by one fill kernel on the current stream (optimizer.zero_grad()); not a memset node: csrc/elementwise.hip"""
    require_device(t)
    if not t.is_contiguous():
        raise ValueError("zero_: contiguous tensor expected")
    check(lib().pcvae_zero(ptr(t), t.numel() * t.element_size(), stream()), "zero")
    return t


# ------------------------------------------------------------------------------------------- K8
def adam_step_(p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0, weight_decay=0.0):
    require_device(p, g, m, v)
    check(lib().pcvae_adam_step_l2(ptr(p, F32), ptr(g, F32), ptr(m, F32), ptr(v, F32), p.numel(), lr, b1, b2, eps, step,
                                   grad_scale, weight_decay, stream()), "adam_step")


# ------------------------------------------------------------- training the click model (pretrain_env.py:25-139)
class _EmbeddingRows(torch.autograd.Function):
    """table[idx] as rows of a [n_idx / group, group * D] matrix; backward = dense table gradient by scatter-add."""

    @staticmethod
    def forward(ctx, table, idx, group):
        out = gather_rows(table, idx, group=group)
        ctx.save_for_backward(idx)
        ctx.group, ctx.shape = group, tuple(table.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = _c2d(g)
        N, D = ctx.shape
        dt = torch.zeros(N, D, dtype=F32, device=g.device)
        check(lib().pcvae_scatter_add_rows(ptr(g, F32), _ld(g), ctx.group, D, ptr(idx), idx.numel(), ptr(dt, F32), N,
                                           stream()), "scatter_add_rows")
        return dt, None, None


def embedding_rows(table, idx, group=1):
    """differentiable nn.Embedding lookup: [idx.numel() / group, group * D]"""
    return _EmbeddingRows.apply(table, idx.reshape(-1).contiguous(), group)


class _NormalizeRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = _c2d(x).clone()
        norm = torch.empty(y.shape[0], dtype=F32, device=y.device)
        check(lib().pcvae_normalize_rows_norm(ptr(y, F32), _ld(y), y.shape[0], y.shape[1], ptr(norm, F32), stream()),
              "normalize_rows_norm")
        ctx.save_for_backward(y, norm)
        return y

    @staticmethod
    def backward(ctx, g):
        y, norm = ctx.saved_tensors
        g = _c2d(g)
        dx = torch.empty_like(y)
        check(lib().pcvae_normalize_rows_bwd(ptr(y, F32), _ld(y), ptr(norm, F32), ptr(g, F32), _ld(g), ptr(dx, F32), _ld(dx),
                                             y.shape[0], y.shape[1], stream()), "normalize_rows_bwd")
        return dx


def normalize_rows(x):
    """F.normalize(x, p=2, dim=1) of a [rows, cols] matrix, differentiable"""
    require_device(x)
    return _NormalizeRows.apply(x)


class _BCESigmoid(torch.autograd.Function):
    """nn.BCELoss()(sigmoid(x), t), mean over all elements (pretrain_env.py:57-58,84)"""

    @staticmethod
    def forward(ctx, x, t):
        x, t = x.contiguous(), t.to(F32).contiguous()
        n = x.numel()
        le = torch.empty(n, dtype=F32, device=x.device)
        dx = torch.empty_like(x)
        check(lib().pcvae_bce_sigmoid(ptr(x, F32), ptr(t, F32), n, ptr(le, F32), ptr(dx, F32), 1.0 / n, stream()), "bce_sigmoid")
        out = torch.empty((), dtype=F32, device=x.device)
        check(lib().pcvae_sum(ptr(le, F32), n, 1.0 / n, ptr(out, F32), stream()), "sum")
        ctx.save_for_backward(dx)
        return out

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        d2 = dx.reshape(1, -1)
        out = torch.empty_like(d2)
        check(lib().pcvae_scale_rows(ptr(d2, F32), _ld(d2), ptr(out, F32), _ld(out), 1, d2.shape[1], ptr(g.contiguous(), F32),
                                     1.0, stream()), "scale_rows")
        return out.reshape(dx.shape), None


def bce_sigmoid(logits, targets):
    require_device(logits, targets)
    return _BCESigmoid.apply(logits, targets)


class _MLPRelu(torch.autograd.Function):
    """Linear -> ReLU -> ... -> Linear (the click model's stack, env/response_model.py:84-86) as one autograd node"""

    @staticmethod
    def forward(ctx, x, *params):
        require_device(x, *params)
        n = len(params) // 2
        h = _c2d(x)
        acts = [h]
        for i in range(n):
            h = linear_fwd_raw(h, params[2 * i], params[2 * i + 1], ACT_RELU if i < n - 1 else ACT_NONE)
            acts.append(h)
        ctx.n = n
        ctx.save_for_backward(*acts, *params)
        return h

    @staticmethod
    def backward(ctx, g):
        n = ctx.n
        saved = ctx.saved_tensors
        acts, params = saved[: n + 1], saved[n + 1:]
        g = _c2d(g).contiguous()
        grads = [None] * (2 * n)
        for i in range(n - 1, -1, -1):
            W, b = params[2 * i], params[2 * i + 1]
            dW, db = torch.zeros_like(W), torch.zeros_like(b)
            linear_bwd_weight_raw(g, acts[i], dW, db)
            grads[2 * i], grads[2 * i + 1] = dW, db
            if i > 0 or ctx.needs_input_grad[0]:
                g = linear_bwd_input_raw(g, W, xact=None)
                if i > 0:  # acts[i] is the ReLU output of layer i-1
                    check(lib().pcvae_relu_bwd(ptr(g, F32), _ld(g), ptr(acts[i], F32), _ld(acts[i]), g.shape[0], g.shape[1],
                                               stream()), "relu_bwd")
            else:
                g = None
        return (g,) + tuple(grads)


def mlp_relu(x, layers):
    flat = []
    for W, b in layers:
        flat += [W, b]
    return _MLPRelu.apply(x, *flat)
