"""Shared helpers for the parity tests."""
import numpy as np
import torch

# north_star: "outputs match the reference PyTorch CPU path ... within 1e-3 relative fp32 per pixel".
# Per-pixel check used throughout:  |got - ref| <= RTOL*|ref| + RTOL*rms(ref)
# (the rms term is the absolute floor that keeps pixels with ref ~ 0 meaningful).
RTOL = 1e-3


def sd_from_npz(arrays, prefix):
    return {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in arrays.items() if k.startswith(prefix)}


def rel_err(got, ref):
    got = torch.as_tensor(got, dtype=torch.float32).cpu()
    ref = torch.as_tensor(ref, dtype=torch.float32).cpu()
    rms = ref.pow(2).mean().sqrt().item() + 1e-12
    return ((got - ref).abs() / (ref.abs() + rms)).max().item()


def assert_close(got, ref, rtol=RTOL, what=""):
    got = torch.as_tensor(got, dtype=torch.float32).cpu()
    ref = torch.as_tensor(ref, dtype=torch.float32).cpu()
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (what, tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got).all(), "%s: non-finite values" % what
    e = rel_err(got, ref)
    assert e <= rtol, "%s: per-pixel relative error %.3e > %.1e (max abs diff %.3e)" % (
        what, e, rtol, (got - ref).abs().max().item())
    return e
