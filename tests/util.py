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


# torchvision.models.vgg19().features, configuration 'E': index -> out channels of the Conv2d(3x3, pad 1) there
VGG19_FEATURE_CONVS = {0: (3, 64), 2: (64, 64), 5: (64, 128), 7: (128, 128), 10: (128, 256), 12: (256, 256),
                       14: (256, 256), 16: (256, 256), 19: (256, 512), 21: (512, 512), 23: (512, 512), 25: (512, 512),
                       28: (512, 512), 30: (512, 512), 32: (512, 512), 34: (512, 512)}


def seeded_vgg19_features(seed=77, upto=30):
    """Deterministic, host-independent stand-in for torchvision's pretrained vgg19 `features` state_dict (a download;
    12.9 M values up to index 30 are too many for a fixture): numpy MT19937, one stream per tensor.  He-scaled uniform
    weights (variance 2/fan_in keeps activations O(1) through 13 layers) and small non-zero biases.
    Returns {'features.<idx>.weight' / '.bias': tensor}."""
    out = {}
    for idx, (cin, cout) in sorted(VGG19_FEATURE_CONVS.items()):
        if idx >= upto:
            continue
        rs = np.random.RandomState(seed * 1000 + idx)
        a = float(np.sqrt(3.0 * 2.0 / (cin * 9)))
        out["features.%d.weight" % idx] = torch.from_numpy(rs.uniform(-a, a, size=(cout, cin, 3, 3)).astype(np.float32))
        out["features.%d.bias" % idx] = torch.from_numpy(rs.uniform(-0.05, 0.05, size=(cout,)).astype(np.float32))
    return out


def vgg19_slice_state_dict(features_sd):
    """torchvision 'features.<idx>.*' keys -> the reference Vgg19's 'slice<k>.<idx>.*' keys (models/networks.py:846-860)."""
    cuts = [(0, 2), (2, 7), (7, 12), (12, 21), (21, 30)]
    out = {}
    for k, v in features_sd.items():
        idx = int(k.split(".")[1])
        for s, (lo, hi) in enumerate(cuts):
            if lo <= idx < hi:
                out["slice%d.%s" % (s + 1, k[len("features."):])] = v
    return out
