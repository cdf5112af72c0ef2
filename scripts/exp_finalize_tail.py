import sys, os
sys.path.insert(0, "/root/repo")
import torch, torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine
eng = Engine("cuda:0", L.BF16)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
    return sorted(ts)[reps//2]
for (cin,cout,k,H,W) in [(108,32,7,1024,2048),(108,16,7,1024,2048),(32,32,3,1024,2048),(64,64,3,512,1024)]:
    mod = nn.Conv2d(cin,cout,k,padding=0).to("cuda:0")
    norm = nn.BatchNorm2d(cout).to("cuda:0")
    x = eng.pack(torch.randn(1,cin,H,W,device="cuda:0"))
    ss = torch.zeros(4*cout, device="cuda:0")
    for tile in (13, 16, 14, 3):
        eng.tile_override[(cin,cout,k,1,0)] = tile
        a = t(lambda: eng.conv(x, mod, L.PAD_REFLECT, k//2, L.OUT_RAW_F32_NHWC, want_stats=True, fin=(norm, ss)))
        b = t(lambda: eng.conv(x, mod, L.PAD_REFLECT, k//2, L.OUT_RAW_F32_NHWC, want_stats=True))
        c = t(lambda: eng.conv(x, mod, L.PAD_REFLECT, k//2, L.OUT_RAW_F32_NHWC, want_stats=False))
        rows = eng.conv_log[-1]
        print("%d->%d k%d @%dx%d tile %d: fused-finalize %.0f us | stats only %.0f us | no stats %.0f us" % (cin,cout,k,H,W,tile,a,b,c), flush=True)
