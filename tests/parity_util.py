"""Error measures shared by the GPU parity tests.

Stated bf16 tolerance of this build (every matrix-product operand is rounded to bf16, 2^-9 relative per rounding;
accumulation, statistics and the residual stream are f32), ELEMENTWISE:

    |got - ref| <= ATOL_RMS * scale(ref) + RTOL * |ref|,   scale = max(rms of the element's row, rms of the tensor)

with RTOL = 1e-2 and ATOL_RMS per tensor class below (a row = the last dimension).  The absolute part is tied to an
RMS because outputs and gradients are sums of many rounded terms: an element that happens to be near zero still
carries the rounding noise of the terms that cancelled in it.  Per-token quantities (attention outputs, input
gradients) are heavy tailed over tokens -- a token whose gradient is 50 times the typical one carries 50 times the
noise in ALL its components -- hence the row RMS; the tensor RMS floors it for rows that are themselves small.
`margin(got, ref)` is the smallest ATOL_RMS that makes the bound hold for a tensor pair; tests print it and assert
it against the class limit (limits are at most twice the largest value measured on MI355X over the test cases of
that class).
The gradients of to_q / to_k are the least well conditioned quantities here (small differences of large terms: their
norm is 50 times below to_v's): flipping the bf16 rounding of one LayerNorm output element in 70,000 moves their
largest element-wise margin by as much as the limit allows, while their relative L2 error stays near 1e-2.
"""
import torch

RTOL = 1e-2
ATOL_RMS_ACT = 4e-2      # logits, encodings, attention / layer outputs   (largest measured: 2.1e-2)
ATOL_RMS_GRAD = 1.6e-1   # parameter and input gradients                  (largest measured: 8.6e-2)
# Whole-tensor summaries of the same tolerance, for tests that compare a model's outputs and gradients tensor by
# tensor (they follow from the elementwise bound for tensors without heavy tails; asserted on their own because a
# uniform drift of 1 % would pass the elementwise bound through its relative part):
REL_MAX_ACT = 6e-3       # max |got - ref| / max |ref| of logits / encodings        (largest measured: 3.1e-3)
REL_L2_ACT = 4e-3        # relative L2 error of logits / encodings                  (largest measured: 1.5e-3)
REL_L2_GRAD = 3e-2       # relative L2 error of a parameter gradient tensor         (largest measured: 1.1e-2)


def _d(t):
    return torch.as_tensor(t).detach().double().cpu()


def relmax(a, b):
    a, b = _d(a), _d(b)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rell2(a, b):
    a, b = _d(a), _d(b)
    return float((a - b).norm() / (b.norm() + 1e-30))


def margin(got, ref, rtol=RTOL):
    """smallest a >= 0 with |got - ref| <= a * max(rms(row), rms(tensor)) + rtol * |ref| elementwise"""
    got, ref = _d(got), _d(ref)
    if ref.dim() == 0:
        got, ref = got.reshape(1), ref.reshape(1)
    rms = float(ref.pow(2).mean().sqrt()) + 1e-30
    scale = ref.pow(2).mean(dim=-1, keepdim=True).sqrt().clamp_min(rms)
    return max(0.0, float((((got - ref).abs() - rtol * ref.abs()) / scale).max()))


def report(name, got, ref):
    m, l2, mx = margin(got, ref), rell2(got, ref), relmax(got, ref)
    print(f"{name}: elementwise margin {m:.3e} x rms, rel L2 {l2:.3e}, max/max {mx:.3e}")
    return m, l2, mx
