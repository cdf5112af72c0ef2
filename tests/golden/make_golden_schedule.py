#!/usr/bin/env python3
"""Call traces of the REFERENCE's resume / schedule helpers (models/models.py:104-163: init_params, save_models,
update_models) and of BaseModel.update_training_batch / update_learning_rate arithmetic (models/base_model.py:154-181) over
a grid of option sets, with recording mock objects.  Build container only.   python tests/golden/make_golden_schedule.py"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG


class Rec:
    def __init__(self, log, name):
        self.log, self.name = log, name
        self.module = self
        self.dataset = self

    def __getattr__(self, attr):
        def call(*a, **k):
            self.log.append([self.name, attr] + [x if isinstance(x, (int, str, float)) else str(x) for x in a])
        return call

    def __len__(self):
        return 37


def scenarios():
    for cont, (ep, it) in ((False, (1, 0)), (True, (3, 5)), (True, (12, 7)), (True, (25, 0))):
        for S, fix in ((1, 0), (2, 10), (3, 5)):
            for batch in (1, 4):
                yield dict(continue_train=cont, iter=(ep, it), n_scales_spatial=S, niter_fix_global=fix, batchSize=batch,
                           niter=10, niter_decay=10, niter_step=5, n_gpus_gen=2, n_frames_G=3, n_frames_D=3, output_nc=3,
                           n_scales_temporal=2, label_nc=35 if S > 1 else 0, input_nc=15, print_freq=100,
                           save_latest_freq=1000, save_epoch_freq=2)


def main():
    MG.install_shims()
    import fractions, math
    fractions.gcd = math.gcd              # removed in Python 3.9; models/models.py:7-8 still calls it
    from models import models as R
    out = []
    for sc in scenarios():
        ck = tempfile.mkdtemp()
        os.makedirs(os.path.join(ck, "x"))
        opt = types.SimpleNamespace(checkpoints_dir=ck, name="x", **{k: v for k, v in sc.items() if k != "iter"})
        if sc["continue_train"]:
            np.savetxt(os.path.join(ck, "x", "iter.txt"), sc["iter"], delimiter=",", fmt="%d")
        log = []
        G, D, loader, vis = Rec(log, "G"), Rec(log, "D"), Rec(log, "loader"), Rec(log, "vis")
        ret = R.init_params(opt, G, D, loader)
        rec = {"scenario": sc, "init_ret": [int(v) for v in ret[:-1]], "init_log": list(log), "epochs": []}
        for epoch in (2, 5, 10, 11, 15, 20):
            del log[:]
            R.update_models(opt, epoch, G, D, loader)
            upd = list(log)
            del log[:]
            R.save_models(opt, epoch, 3, 2000, vis, ret[-1], G, D, end_of_epoch=False)
            R.save_models(opt, epoch, 3, 2001, vis, ret[-1], G, D, end_of_epoch=True)
            itxt = open(ret[-1]).read().split() if os.path.exists(ret[-1]) else None
            rec["epochs"].append({"epoch": epoch, "update_log": upd, "save_log": [l for l in log if l[0] != "vis"], "iter_txt": itxt})
        out.append(rec)
    path = os.path.join(HERE, "schedule_traces.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, os.path.getsize(path) // 1024, "KB", len(out), "scenarios")


if __name__ == "__main__":
    main()
