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


def seeded_flownet2_weights(shapes, seed=2024, flow_head_scale=1.0):
    """Deterministic, host-independent FlowNet2 weights (162.5 M values are too many to store in a fixture):
    numpy's MT19937 RandomState, one stream per tensor keyed by crc32(name).  Xavier-uniform weights and
    U(0, 0.1) biases as models/flownet2_pytorch/models.py:68-77 draws them (biases scaled down), with the
    predict_flow heads scaled so that random-init flows stay a few pixels.
    shapes: {state_dict key: shape}.  Returns {key: torch.float32 tensor}."""
    import zlib
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        rs = np.random.RandomState((seed + zlib.crc32(name.encode())) & 0x7fffffff)
        if name.endswith(".weight"):
            rf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
            fan_in, fan_out = shp[1] * rf, shp[0] * rf
            a = float(np.sqrt(6.0 / (fan_in + fan_out)))
            w = rs.uniform(-a, a, size=shp).astype(np.float32)
            if "predict_flow" in name:
                w *= flow_head_scale
        else:
            w = rs.uniform(0.0, 0.1, size=shp).astype(np.float32)
            if "predict_flow" in name:
                w *= flow_head_scale
        out[name] = torch.from_numpy(w)
    return out
