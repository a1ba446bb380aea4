import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def has_gpu() -> bool:
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def lib():
    """libcid.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from consistentid_amd import _lib, build
    build.build(verbose=False)
    return _lib.load()


@pytest.fixture(scope="session")
def dev(lib):
    if not has_gpu():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# fp16 tolerance of the hot path (BASELINE.json north_star: "within 1e-3 rel-fp16"):
# relative L2 error of a kernel's fp16 output against the fp32/fp64 oracle on identical
# fp16-representable inputs.  max-abs error is bounded at 4e-3 of the output range.
TOL_L2 = 1e-3
TOL_MAX = 4e-3


def check_close(got, ref, what="", tol_l2=TOL_L2, tol_max=TOL_MAX):
    e2, em = rel_l2(got, ref), max_rel(got, ref)
    print(f"[parity] {what}: rel_l2={e2:.3e} max_rel={em:.3e}")
    assert torch.isfinite(got.float()).all(), f"{what}: non-finite output"
    assert e2 <= tol_l2 and em <= tol_max, f"{what}: rel_l2={e2:.3e} (tol {tol_l2}) max_rel={em:.3e} (tol {tol_max})"
    return e2, em


def check_vs_fp16_arm(got, ref32, arm16, what="", slack=1.5):
    """Attention-type paths chain several contractions through fp16 intermediates (q, k, v, p, o),
    exactly like the reference's own fp16 pipeline does.  Criterion: our error against the fp32
    oracle must not exceed max(TOL_L2, slack x the error of the SAME computation done with stock
    PyTorch fp16 ops on the GPU) -- i.e. we are at least as accurate as the reference-precision arm."""
    e_arm, m_arm = rel_l2(arm16, ref32), max_rel(arm16, ref32)
    e2, em = rel_l2(got, ref32), max_rel(got, ref32)
    tol = max(TOL_L2, slack * e_arm)
    # the largest single error is held to the arm's as well: a few wrong rows / columns of a tiled kernel barely move
    # the L2 norm of a large tensor
    tol_m = max(TOL_MAX, slack * m_arm)
    print(f"[parity] {what}: rel_l2={e2:.3e} max_rel={em:.3e} | torch-fp16 arm rel_l2={e_arm:.3e} max_rel={m_arm:.3e} "
          f"-> tol {tol:.3e} / {tol_m:.3e}")
    assert torch.isfinite(got.float()).all(), f"{what}: non-finite output"
    assert e2 <= tol, f"{what}: rel_l2={e2:.3e} > {tol:.3e} (fp16 arm {e_arm:.3e})"
    assert em <= tol_m, f"{what}: max_rel={em:.3e} > {tol_m:.3e} (fp16 arm {m_arm:.3e})"
    return e2, e_arm


def half_arm(module, dev):
    """The reference-precision arm of a parity test: a copy of an oracle module (the plain-PyTorch restatement of the
    reference path) in fp16 on the GPU, i.e. what the reference's own `torch_dtype=float16` pipeline computes with stock
    PyTorch-ROCm kernels.  Used with check_vs_fp16_arm: our error against the fp32 oracle may not exceed
    max(TOL_L2, 1.5 x the arm's error)."""
    import copy
    return copy.deepcopy(module).to(dev).half().eval()


def dev_half(x, dev):
    """tensors (also inside lists / dicts) -> fp16 on the GPU; integer tensors and non-tensors unchanged"""
    if isinstance(x, torch.Tensor):
        return x.to(dev).half() if x.is_floating_point() else x.to(dev)
    if isinstance(x, (list, tuple)):
        return type(x)(dev_half(v, dev) for v in x)
    if isinstance(x, dict):
        return {k: dev_half(v, dev) for k, v in x.items()}
    return x
