"""CPU checks of the C-ABI boundary: the shared library loads without a GPU, exports every symbol that
include/scmomer_hip.h declares with the declared number of arguments, and rejects bad arguments with an error
code and a message instead of launching anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "scmomer_hip.h")


def _declared():
    """name -> number of parameters, parsed from the header's prototypes"""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(scm_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_header_parses():
    d = _declared()
    assert len(d) >= 20 and "scm_favor_fwd" in d and "scm_gemm_nt" in d


def test_library_exports_every_declared_symbol():
    from scmomer_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    raw = C.CDLL(_lib.LIB_PATH)
    declared = _declared()
    for name, nargs in declared.items():
        assert hasattr(raw, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
        assert len(_lib.SIGNATURES[name][1]) == nargs, f"{name}: binding has {len(_lib.SIGNATURES[name][1])} args, header {nargs}"
    assert set(_lib.SIGNATURES) == set(declared), set(_lib.SIGNATURES) ^ set(declared)


def test_version_and_host_only_calls():
    from scmomer_amd._lib import FavorPlan, lib
    assert lib.scm_version() >= 1
    plan = FavorPlan()
    assert lib.scm_favor_make_plan(3, 10, 16906, C.byref(plan)) == 0
    assert (plan.B, plan.H, plan.N) == (3, 10, 16906)
    assert plan.kmax_chunks >= 1 and plan.ctx_chunks >= 1 and plan.apply_chunks >= 1
    assert lib.scm_favor_ws_floats(C.byref(plan)) > 0
    assert lib.scm_wgrad_ws_floats(1920, 21) > 0
    assert lib.scm_ln_bwd_ws_floats(50718, 200) > 0
    n = lib.scm_trace_kernel_count()
    names = [lib.scm_trace_kernel_name(i).decode() for i in range(n)]
    assert "favor_apply_fwd" in names and "gemm_skinny/bf16" in names and len(set(names)) == n


def test_bad_arguments_are_rejected_without_a_launch():
    from scmomer_amd._lib import ScmError, check, lib
    rc = lib.scm_gemm_nt(0, 0, None, 200, None, 200, 128, 640, 200, None, 640, None, 0, None, None, 0, None, 0, None)
    assert rc < 0
    assert b"null" in lib.scm_last_error()
    rc = lib.scm_favor_make_plan(0, 10, 100, None)
    assert rc < 0
    with pytest.raises(ScmError):
        check(lib.scm_ln_fwd(None, None, None, 10, 200, None, None, None, None), "scm_ln_fwd")
    # sizes the kernels do not support: K not a multiple of 8
    buf = C.create_string_buffer(64)
    addr = C.addressof(buf)
    addr += (-addr) % 16
    rc = lib.scm_gemm_nt(0, 0, addr, 8, addr, 8, 1, 8, 4, addr, 8, None, 0, None, None, 0, None, 0, None)
    assert rc < 0 and b"K%8" in lib.scm_last_error()


def test_product_code_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under scmomer_amd/ may import it"""
    pkg = os.path.join(ROOT, "scmomer_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "oracle/" not in text and "performer_oracle" not in text, f


def test_plan_fills_whole_rounds_of_workgroups():
    """chunks per (cell, head) are chosen so that the attention launches are a nearly whole number of rounds of 256
    workgroups; the key-stabiliser partial count follows the context chunks (9 waves each)"""
    from scmomer_amd._lib import FavorPlan, lib
    for b, h, n in [(1, 10, 16906), (3, 10, 16906), (16, 10, 16906), (24, 10, 16906), (64, 10, 16906), (2, 3, 131),
                    (1, 1, 40)]:
        plan = FavorPlan()
        assert lib.scm_favor_make_plan(b, h, n, C.byref(plan)) == 0
        assert plan.kmax_chunks == 9 * plan.ctx_chunks
        assert 1 <= plan.apply_chunks <= (n + 31) // 32 and 1 <= plan.ctx_chunks <= (n + 63) // 64
        if n == 16906:
            rounds = b * h * plan.apply_chunks / 256
            assert rounds / -(-rounds // 1) >= 0.9, (b, plan.apply_chunks)
    plan = FavorPlan()
    lib.scm_favor_make_plan(3, 10, 16906, C.byref(plan))
    assert (plan.apply_chunks, plan.ctx_chunks) == (8, 8)          # the benchmark configuration: one round
