"""ctypes loader of oracle/_ref/libref_ops.so: the reference's OWN Correlation / Resample2d / ChannelNorm CUDA kernels
(models/flownet2_pytorch/networks/*_package/*.cu) executed on host cores by oracle/ref_ops/cuda_emu.h.

TEST INFRASTRUCTURE: imported by tests/ only.  This is what pins the two restatements of these ops
(oracle/vid2vid_oracle.py, oracle/native_ops_scalar.py) and the HIP kernels to reference CODE rather than to a reading of
it.  Built by oracle/ref_ops/build.sh (run by __graft_entry__.build() where /root/reference exists); the library is
git-ignored and travels to the GPU box with the snapshot.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libref_ops.so")
_lib = None


def available(build=True):
    """True when the library can be loaded; builds it first when the reference tree is present."""
    if not os.path.exists(SO) and build and os.path.isdir(os.environ.get("V2V_REFERENCE", "/root/reference")):
        subprocess.check_call(["bash", os.path.join(_HERE, "ref_ops", "build.sh")])
    return os.path.exists(SO)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libref_ops.so is missing: run oracle/ref_ops/build.sh where /root/reference exists")
        _lib = C.CDLL(SO)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        _lib.ref_correlation_out_size.argtypes = [C.c_int] * 7 + [ip, ip, ip]
        _lib.ref_correlation_forward.argtypes = [fp, fp, fp] + [C.c_int] * 9
        _lib.ref_resample2d_forward.argtypes = [fp, fp, fp] + [C.c_int] * 7
        _lib.ref_resample2d_backward.argtypes = [fp] * 5 + [C.c_int] * 7
        _lib.ref_channelnorm_forward.argtypes = [fp, fp] + [C.c_int] * 5
        _lib.ref_channelnorm_backward.argtypes = [fp] * 4 + [C.c_int] * 5
    return _lib


def _p(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu"
    return C.cast(t.data_ptr(), C.POINTER(C.c_float))


def _c(t):
    return t.detach().float().cpu().contiguous()


def correlation(in1, in2, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2):
    """correlation_cuda.forward (correlation_cuda.cc:10-87) -> (N, D*D, oh, ow)."""
    in1, in2 = _c(in1), _c(in2)
    N, Cc, H, W = in1.shape
    oc, oh, ow = C.c_int(), C.c_int(), C.c_int()
    lib().ref_correlation_out_size(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, oc, oh, ow)
    out = torch.empty(N, oc.value, oh.value, ow.value)
    rc = lib().ref_correlation_forward(_p(in1), _p(in2), _p(out), N, Cc, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    assert rc == 1
    return out


def resample2d(img, flow, kernel_size=1):
    """resample2d_cuda.forward (resample2d_cuda.cc:6-13); output has the flow's spatial size (resample2d.py:14-17)."""
    img, flow = _c(img), _c(flow)
    N, Cc, H, W = img.shape
    _, _, OH, OW = flow.shape
    out = torch.empty(N, Cc, OH, OW)
    lib().ref_resample2d_forward(_p(img), _p(flow), _p(out), N, Cc, H, W, OH, OW, kernel_size)
    return out


def resample2d_backward(img, flow, grad_out, kernel_size=1):
    img, flow, grad_out = _c(img), _c(flow), _c(grad_out)
    N, Cc, H, W = img.shape
    _, _, OH, OW = flow.shape
    g_img, g_flow = torch.empty_like(img), torch.empty_like(flow)
    lib().ref_resample2d_backward(_p(img), _p(flow), _p(grad_out), _p(g_img), _p(g_flow), N, Cc, H, W, OH, OW, kernel_size)
    return g_img, g_flow


def channelnorm(x, norm_deg=2):
    x = _c(x)
    N, Cc, H, W = x.shape
    out = torch.empty(N, 1, H, W)
    lib().ref_channelnorm_forward(_p(x), _p(out), N, Cc, H, W, norm_deg)
    return out


def channelnorm_backward(x, out, grad_out, norm_deg=2):
    x, out, grad_out = _c(x), _c(out), _c(grad_out)
    N, Cc, H, W = x.shape
    g = torch.empty_like(x)
    lib().ref_channelnorm_backward(_p(x), _p(out), _p(grad_out), _p(g), N, Cc, H, W, norm_deg)
    return g
