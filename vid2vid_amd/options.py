"""Flat option namespace with the reference's defaults for the hot-path flags.

The reference's argparse front end (options/base_options.py:11-85, train_options.py:5-43,
test_options.py:5-15) is kept verbatim by a drop-in user; this helper only builds the same
namespace programmatically for tests, bench.py and smoke() -- names, defaults and the
post-processing of `parse()` (base_options.py:96-109) are the reference's.
"""
from types import SimpleNamespace

_DEFAULTS = dict(
    # base_options.py
    dataroot="datasets/Cityscapes/", batchSize=1, loadSize=512, fineSize=512, input_nc=3, label_nc=0, output_nc=3,
    netG="composite", ngf=128, ndf=64, n_blocks=9, n_downsample_G=3, gpu_ids=[0], n_gpus_gen=-1,
    name="experiment_name", dataset_mode="temporal", model="vid2vid", checkpoints_dir="./checkpoints", norm="batch",
    use_instance=False, label_feat=False, feat_num=3, n_blocks_local=3, n_local_enhancers=1,
    n_frames_G=3, n_scales_spatial=1, no_first_img=False, use_single_G=False, fg=False, fg_labels=[26], no_flow=False,
    openpose_only=False, densepose_only=False, add_face_disc=False, load_pretrain="", debug=False, fp16=False,
    local_rank=0,
    # test_options.py
    which_epoch="latest", use_real_img=False, how_many=300,
    # train_options.py
    isTrain=False, continue_train=False, niter=10, niter_decay=10, beta1=0.5, lr=0.0002, TTUR=False,
    gan_mode="ls", num_D=2, pool_size=1, n_layers_D=3, no_vgg=False, no_ganFeat=False, lambda_feat=10.0, lambda_F=10.0,
    lambda_T=10.0, sparse_D=False, n_scales_temporal=2, n_frames_D=3, n_frames_total=30, max_frames_per_gpu=1,
    max_frames_backpropagate=1, max_t_step=1, niter_step=5, niter_fix_global=0,
    # vid2vid_amd additions
    precision=None,           # 'fp32' | 'bf16' | None (None: bf16 iff opt.fp16)
    random_init_ok=False,     # allow create_model() without a G0 checkpoint (benchmarks / smoke)
    vgg19_checkpoint="checkpoints/vgg19-dcbb9e9d.pth",   # torchvision's vgg19 state_dict (the reference downloads it)
    fix_update_fixed_params=False,   # True: update_fixed_params rebuilds the captured optimizer over all scales (the reference's
                                     # evident intent); False: the reference's observable behaviour (models/base_model.py docstring)
)


def make_opt(**overrides):
    d = dict(_DEFAULTS)
    unknown = set(overrides) - set(d)
    if unknown:
        raise KeyError("unknown option(s): %s" % sorted(unknown))
    d.update(overrides)
    opt = SimpleNamespace(**d)
    if isinstance(opt.fg_labels, str):
        opt.fg_labels = [int(s) for s in opt.fg_labels.split(",") if int(s) >= 0]
    if isinstance(opt.gpu_ids, str):
        opt.gpu_ids = [int(s) for s in opt.gpu_ids.split(",") if int(s) >= 0]
    if opt.n_gpus_gen == -1:
        opt.n_gpus_gen = len(opt.gpu_ids)
    return opt
