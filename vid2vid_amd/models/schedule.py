"""Resume / checkpoint / schedule helpers that train.py imports from models.models
(reference models/models.py:104-163).  Pure host logic, same return shapes and file formats
(`iter.txt` = "epoch,iter"; `<epoch>_net_<label>.pth`)."""
import math
import os

import numpy as np


def init_params(opt, modelG, modelD, data_loader):
    iter_path = os.path.join(opt.checkpoints_dir, opt.name, "iter.txt")
    start_epoch, epoch_iter = 1, 0
    if opt.continue_train:
        if os.path.exists(iter_path):
            start_epoch, epoch_iter = [int(v) for v in np.loadtxt(iter_path, delimiter=",", dtype=int)]
        print("Resuming from epoch %d at iteration %d" % (start_epoch, epoch_iter))
        if start_epoch > opt.niter:
            modelG.module.update_learning_rate(start_epoch - 1, "G")
            modelD.module.update_learning_rate(start_epoch - 1, "D")
        if opt.n_scales_spatial > 1 and opt.niter_fix_global != 0 and start_epoch > opt.niter_fix_global:
            modelG.module.update_fixed_params()
        if start_epoch > opt.niter_step:
            ratio = (start_epoch - 1) // opt.niter_step
            data_loader.dataset.update_training_batch(ratio)
            modelG.module.update_training_batch(ratio)

    n_gpus = opt.n_gpus_gen if opt.batchSize == 1 else 1
    tG, tD = opt.n_frames_G, opt.n_frames_D
    tDB = tD * opt.output_nc
    s_scales, t_scales = opt.n_scales_spatial, opt.n_scales_temporal
    input_nc = 1 if opt.label_nc != 0 else opt.input_nc
    output_nc = opt.output_nc
    print_freq = opt.print_freq * opt.batchSize // math.gcd(opt.print_freq, opt.batchSize)   # lcm
    total_steps = (start_epoch - 1) * len(data_loader) + epoch_iter
    total_steps = total_steps // print_freq * print_freq
    return (n_gpus, tG, tD, tDB, s_scales, t_scales, input_nc, output_nc, start_epoch, epoch_iter, print_freq,
            total_steps, iter_path)


def save_models(opt, epoch, epoch_iter, total_steps, visualizer, iter_path, modelG, modelD, end_of_epoch=False):
    if not end_of_epoch:
        if total_steps % opt.save_latest_freq == 0:
            visualizer.vis_print("saving the latest model (epoch %d, total_steps %d)" % (epoch, total_steps))
            modelG.module.save("latest")
            modelD.module.save("latest")
            np.savetxt(iter_path, (epoch, epoch_iter), delimiter=",", fmt="%d")
    elif epoch % opt.save_epoch_freq == 0:
        visualizer.vis_print("saving the model at the end of epoch %d, iters %d" % (epoch, total_steps))
        for label in ("latest", epoch):
            modelG.module.save(label)
            modelD.module.save(label)
        np.savetxt(iter_path, (epoch + 1, 0), delimiter=",", fmt="%d")


def update_models(opt, epoch, modelG, modelD, data_loader):
    if epoch > opt.niter:                                   # linear lr decay
        modelG.module.update_learning_rate(epoch, "G")
        modelD.module.update_learning_rate(epoch, "D")
    if (epoch % opt.niter_step) == 0:                       # grow the training sequence length
        data_loader.dataset.update_training_batch(epoch // opt.niter_step)
        modelG.module.update_training_batch(epoch // opt.niter_step)
    if opt.n_scales_spatial > 1 and opt.niter_fix_global != 0 and epoch == opt.niter_fix_global:
        modelG.module.update_fixed_params()                  # start finetuning all scales
