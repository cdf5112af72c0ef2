"""On-GPU counterparts of the reference's util/util.py tensor -> image helpers.

The reference's test.py (:43-54) converts every generated frame on the host: a D2H copy of the fp32 planes (36 x H x W floats for
the label map alone), then numpy (`util.tensor2label`, `util.tensor2im`), then PIL.  Here the conversion is a HIP kernel on the
device and only the uint8 HWC image crosses PCIe; `AsyncImageWriter` encodes / writes files on a worker thread so the save path
does not stall a generator that runs at hundreds of frames per second.

    tensor2im(t, normalize=True)  -> uint8 device tensor (H, W, C) or (H, W) for one plane      (util/util.py:48-71)
    tensor2label(t, n_label)      -> uint8 device tensor (H, W, 3)                              (util/util.py:73-87)
    tensor2flow(t)                -> uint8 device tensor (H, W, 3), HSV-coded flow picture      (util/util.py:89-107)
    to_numpy(img)                 -> what the reference's function returns (numpy uint8)
"""
import ctypes as C
import queue
import threading

import numpy as np
import torch

from .lib import lib, check

# Cityscapes palettes of util/util.py:156-168 (label ids -> RGB; the public Cityscapes colour coding)
_CITY35 = [(0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0), (111, 74, 0), (81, 0, 81), (128, 64, 128), (244, 35, 232),
           (250, 170, 160), (230, 150, 140), (70, 70, 70), (102, 102, 156), (190, 153, 153), (180, 165, 180), (150, 100, 100),
           (150, 120, 90), (153, 153, 153), (153, 153, 153), (250, 170, 30), (220, 220, 0), (107, 142, 35), (152, 251, 152),
           (70, 130, 180), (220, 20, 60), (255, 0, 0), (0, 0, 142), (0, 0, 70), (0, 60, 100), (0, 0, 90), (0, 0, 110),
           (0, 80, 100), (0, 0, 230), (119, 11, 32), (0, 0, 142)]
_CITY20 = [(128, 64, 128), (244, 35, 232), (70, 70, 70), (102, 102, 156), (190, 153, 153), (153, 153, 153), (250, 170, 30),
           (220, 220, 0), (107, 142, 35), (152, 251, 152), (70, 130, 180), (220, 20, 60), (255, 0, 0), (0, 0, 142), (0, 0, 70),
           (0, 60, 100), (0, 80, 100), (0, 0, 230), (119, 11, 32), (0, 0, 0)]


def labelcolormap(n):
    """util/util.py:156-181: the two Cityscapes tables, otherwise the bit-interleaved PASCAL-style map."""
    if n == 35:
        return np.array(_CITY35, dtype=np.uint8)
    if n == 20:
        return np.array(_CITY20, dtype=np.uint8)
    cmap = np.zeros((n, 3), dtype=np.uint8)
    for i in range(n):
        r = g = b = 0
        idx = i
        for j in range(7):
            r ^= (idx & 1) << (7 - j)
            g ^= ((idx >> 1) & 1) << (7 - j)
            b ^= ((idx >> 2) & 1) << (7 - j)
            idx >>= 3
        cmap[i] = (r, g, b)
    return cmap


_CMAPS = {}


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else None


def _planes(t):
    """The reference's leading-dim rules (util/util.py:57-60): 5-D -> [0, -1], 4-D -> [0]."""
    if t.dim() == 5:
        t = t[0, -1]
    if t.dim() == 4:
        t = t[0]
    return t.detach().float().contiguous()


def tensor2im(image_tensor, normalize=True):
    if isinstance(image_tensor, list):
        return [tensor2im(t, normalize) for t in image_tensor]
    x = _planes(image_tensor)[:3].contiguous()
    Cc, H, W = x.shape
    out = torch.empty((H, W, Cc), dtype=torch.uint8, device=x.device)
    check(lib.v2v_tensor2im(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), Cc, H, W, int(bool(normalize)), _stream(x)),
          "tensor2im")
    return out[:, :, 0] if Cc == 1 else out


def tensor2label(output, n_label):
    x = _planes(output)
    Cc, H, W = x.shape
    key = (n_label, str(x.device))
    if key not in _CMAPS:
        _CMAPS[key] = torch.from_numpy(labelcolormap(n_label)[:n_label].copy()).to(x.device)
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=x.device)
    check(lib.v2v_tensor2label(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(_CMAPS[key].data_ptr()),
                               n_label, Cc, H, W, _stream(x)), "tensor2label")
    return out


def tensor2flow(output):
    """util/util.py:89-107 (used by save_all_tensors :38-41 for `flow_ref` and `flow`): direction -> hue, magnitude (min-max
    normalised over the image) -> value.  The reference computes it with OpenCV on the host after a D2H copy of the flow."""
    x = _planes(output)
    if x.shape[0] != 2:
        raise ValueError("tensor2flow expects 2 flow planes, got %d" % x.shape[0])
    _, H, W = x.shape
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=x.device)
    ws = torch.empty(2, dtype=torch.int32, device=x.device)
    check(lib.v2v_tensor2flow(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), H, W, _stream(x)),
          "tensor2flow")
    return out


def to_numpy(img):
    return img.cpu().numpy()


class AsyncImageWriter:
    """Saves uint8 HWC device images without stalling the caller: the D2H copy goes to a pinned staging buffer on a side
    stream, file encoding / writing (PIL, as util.save_image :127-129) happens on a worker thread.
        w = AsyncImageWriter(); w.save(tensor2im(fake_B), "out/frame0001.jpg"); ...; w.close()"""

    def __init__(self, depth=8):
        self.q = queue.Queue(maxsize=depth)
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.errors = []
        self.worker = threading.Thread(target=self._run, daemon=True)
        self.worker.start()

    def _run(self):
        from PIL import Image
        while True:
            item = self.q.get()
            if item is None:
                return
            host, event, path = item
            try:
                if event is not None:
                    event.synchronize()
                Image.fromarray(host.numpy()).save(path)
            except Exception as ex:          # surfaced by close()
                self.errors.append((path, ex))

    def save(self, img, path):
        if img.is_cuda:
            host = torch.empty(img.shape, dtype=torch.uint8, pin_memory=True)
            self.stream.wait_stream(torch.cuda.current_stream(img.device))
            with torch.cuda.stream(self.stream):
                host.copy_(img, non_blocking=True)
                event = torch.cuda.Event()
                event.record(self.stream)
            img.record_stream(self.stream)
        else:
            host, event = img.clone(), None
        self.q.put((host, event, path))

    def close(self):
        self.q.put(None)
        self.worker.join()
        if self.errors:
            raise RuntimeError("image writer failed: %s" % self.errors[:3])
