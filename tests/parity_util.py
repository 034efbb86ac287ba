"""Error measures shared by the GPU parity tests.

Stated bf16 tolerance of this build (every matrix-product operand is rounded to bf16, 2^-9 relative per rounding;
accumulation, statistics and the residual stream are f32), ELEMENTWISE:

    |got - ref| <= ATOL_RMS * rms(ref) + RTOL * |ref|

with RTOL = 1e-2 and ATOL_RMS per tensor class below.  The absolute part is tied to the tensor's own RMS because
outputs and gradients are sums of many rounded terms: an element that happens to be near zero still carries the
rounding noise of the terms that cancelled in it.  `margin(got, ref)` is the smallest ATOL_RMS that makes the bound
hold for a tensor pair; tests print it and assert it against the class limit (limits are at most twice the largest
value measured on MI355X over the test cases of that class).
"""
import torch

RTOL = 1e-2
ATOL_RMS_ACT = 2e-2      # logits, encodings, attention / layer outputs
ATOL_RMS_GRAD = 8e-2     # parameter and input gradients


def _d(t):
    return torch.as_tensor(t).detach().double().cpu()


def relmax(a, b):
    a, b = _d(a), _d(b)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rell2(a, b):
    a, b = _d(a), _d(b)
    return float((a - b).norm() / (b.norm() + 1e-30))


def margin(got, ref, rtol=RTOL):
    """smallest a >= 0 with |got - ref| <= a * rms(ref) + rtol * |ref| elementwise"""
    got, ref = _d(got), _d(ref)
    rms = float(ref.pow(2).mean().sqrt()) + 1e-30
    return max(0.0, float(((got - ref).abs() - rtol * ref.abs()).max()) / rms)


def report(name, got, ref):
    m, l2, mx = margin(got, ref), rell2(got, ref), relmax(got, ref)
    print(f"{name}: elementwise margin {m:.3e} x rms, rel L2 {l2:.3e}, max/max {mx:.3e}")
    return m, l2, mx
