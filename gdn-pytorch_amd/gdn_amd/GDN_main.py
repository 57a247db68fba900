"""Entry point with the reference's CLI: ``python -m gdn_amd.GDN_main DATA --mode {DtoD,RtoD,...}``.

Mirrors GDN_main.py:22-307 of the reference for the hot path: same flags
(option.py), same mode dispatch (:150-201), same optimiser settings
(Adam, betas from --momentum/--beta, eps 1e-8, weight decay hard-wired to 5e-4,
:157,173).  ``--gpu_num 0,1,2,3`` trains on the listed GPUs from one command like the
reference (README.md:82): ``main`` starts one child process per device
(``distributed.launch_ranks``) and gradients are all-reduced with RCCL
(nn.DataParallel is gone); under ``python -m torch.distributed.run`` each rank
takes its LOCAL_RANK GPU instead.

The KITTI/NYU file pipeline (datasets_list.py / transform_list.py) is host I/O
outside the hot path: pass ``--synthetic`` for KITTI-shaped random batches, or
call ``run(args, train_loader, val_loader)`` with loaders that yield the
reference's sample contract ``(gt[B,1,H,W], rgb[B,3,H,W], sparse[B,1,H,W])`` in [-1,1].
"""
import os
import sys

import torch

from . import distributed as D
from . import option
from .AE_model_unet import AutoEncoder, AutoEncoder_2, AutoEncoder_DtoD
from .optim import Adam
from .synthetic import SyntheticLoader
from .trainer import load_checkpoint, train_AE_DtoD, train_AE_RtoD, validate


def _make_optimizer(model, args):
    return Adam(model.parameters(), args.lr, [args.momentum, args.beta], eps=1e-08, weight_decay=5e-4)


def run(args, train_loader=None, val_loader=None):
    rank, local_rank, world = D.env_rank()
    if world == 1 and "HIP_VISIBLE_DEVICES" not in os.environ and not torch.cuda.is_initialized():
        os.environ["HIP_VISIBLE_DEVICES"] = args.gpu_num.split(",")[0]   # reference: CUDA_VISIBLE_DEVICES=--gpu_num
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: the MI355X build has no CPU fallback")
    rank, local_rank, world = D.init()
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(args.seed)          # identical init on every rank
    from . import utils as _U
    _U.GLOBAL_BERHU = bool(getattr(args, "global_berhu", False))
    H, W = args.height, args.width
    if rank == 0:
        print('=> number of GPU processes: ', world)
        print("=> creating model")
    if train_loader is None:
        from .datasets import GpuAugmentLoader, SequenceFolder, SyntheticRawKitti
        steps = args.epoch_size or 100
        if not args.synthetic and os.path.isdir(str(args.data)):
            # the reference's file layout (datasets_list.py:61-76); decode on the host, augment on the GPU
            if args.dataset != "KITTI":
                raise RuntimeError("only the KITTI pipeline (GDN_main.py:56-80) is implemented; NYU is out of scope")
            train_set = SequenceFolder(args.data, args, seed=args.seed, train=True, mode=args.mode)     # same file order on every rank
            val_set = SequenceFolder(args.data, args, seed=args.seed, train=False, mode=args.mode)
            # data parallelism: a common shuffle, rank r takes samples r, r + world, ... (each sample once per epoch);
            # --batch_size is per GPU, so the global batch is world * batch_size at the given learning rate
            train_loader = GpuAugmentLoader(train_set, args.batch_size, dev, train=True, seed=args.seed + rank,
                                            workers=args.workers, drop_last=True, rank=rank, world=world,
                                            order_seed=args.seed + 1)
            val_loader = GpuAugmentLoader(val_set, args.batch_size, dev, train=False, workers=args.workers)
        elif not args.synthetic:
            raise RuntimeError("dataset directory %r not found; pass a KITTI root laid out like the reference's "
                               "(train.txt, val.txt, <scene>/*.jpg, color_gt2/, gt/) or --synthetic" % (args.data,))
        elif getattr(args, "augment", False):
            # synthetic RAW uint8 samples through the same GPU augmentation the file pipeline uses
            raw = SyntheticRawKitti(args.batch_size * min(steps, 8), H, W, seed=args.seed + rank)
            train_loader = GpuAugmentLoader(raw, args.batch_size, dev, train=True, seed=args.seed + rank, drop_last=True)
            val_loader = GpuAugmentLoader(SyntheticRawKitti(args.batch_size * 2, H, W, seed=args.seed + 1000 + rank),
                                          args.batch_size, dev, train=False)
        else:
            train_loader = SyntheticLoader(args.batch_size, steps, H, W, seed=args.seed + rank, device=dev)
            val_loader = SyntheticLoader(args.batch_size, 2, H, W, seed=args.seed + 1000 + rank, device=dev)
    if args.epoch_size == 0:
        args.epoch_size = len(train_loader)
    args.local_rank = local_rank
    logger = object() if args.evaluate else None     # reference: validation only when --evaluate

    if args.mode == 'DtoD':
        G = AutoEncoder_DtoD(norm=args.norm, input_dim=1, height=H, width=W).to(dev).compute_dtype(args.dtype)
        D.broadcast_parameters(G)             # rank 0's weights / BN buffers everywhere (identical seeds make this a no-op)
        opt = _make_optimizer(G, args)
        loss = train_AE_DtoD(args, G, None, None, opt, train_loader, val_loader, args.batch_size, args.epochs,
                             args.lr, logger, None)
        if rank == 0 and loss is not None:
            print('Final loss:', loss.item())
        return loss
    if args.mode in ('RtoD', 'RtoD_single'):
        G = None
        if args.mode == 'RtoD':
            G = AutoEncoder_DtoD(norm=args.norm, input_dim=1, height=H, width=W).to(dev).compute_dtype(args.dtype)
            if os.path.exists(args.model_dir):
                load_checkpoint(G, args.model_dir)
            elif rank == 0:
                print("=> no guide checkpoint at %s: using a randomly initialised guide" % args.model_dir)
            G.eval()
            if getattr(args, "latent_grad", False):
                G.requires_grad_(False)       # the guide only passes d(latent)/d(outputs) through
        R = AutoEncoder_2(norm=args.norm, input_dim=3, height=H, width=W).to(dev).compute_dtype(args.dtype)
        D.broadcast_parameters(R)
        opt = _make_optimizer(R, args)
        return train_AE_RtoD(args, R, G, None, None, opt, train_loader, val_loader, args.batch_size, args.epochs,
                             args.lr, logger, None)
    if args.mode in ('DtoD_test', 'RtoD_test'):
        if args.mode == 'DtoD_test':
            model = AutoEncoder_DtoD(norm=args.norm, input_dim=1, height=H, width=W).to(dev).compute_dtype(args.dtype)
            ckpt = args.model_dir
        else:
            model = AutoEncoder(norm=args.norm, height=H, width=W).to(dev).compute_dtype(args.dtype)
            ckpt = args.RtoD_model_dir
        if os.path.exists(ckpt):
            load_checkpoint(model, ckpt)
        model.eval()
        errors, min_errors, names = validate(args, val_loader, model, 0, logger, args.mode)
        if rank == 0:
            print("Results: " + ", ".join("%s %.4f" % (n, e) for n, e in zip(names, errors)))
        return errors
    raise ValueError("unknown --mode %r" % args.mode)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = option.parse_args(argv)
    devices = [d for d in str(args.gpu_num).split(",") if d != ""]
    if len(devices) > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # the reference's `--gpu_num 0,1,2,3` (README.md:82): one process per listed GPU, started before anything here
        # touches the GPU; this parent only waits and propagates a failure
        rc = D.launch_ranks(argv, devices)
        if rc != 0:
            sys.exit(rc)
        return None
    return run(args)


if __name__ == '__main__':
    main(sys.argv[1:])
