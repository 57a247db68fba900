"""Command-line flags: every flag name and default of the reference's option.py:5-48,
plus the few the MI355X build adds (--synthetic, --local_rank, --dtype, --global_berhu).

Unlike the reference the parser is not evaluated at import time; call ``parse_args()``.
"""
import argparse


def build_parser():
    p = argparse.ArgumentParser(description='Depth AutoEncoder training on KITTI (MI355X-native hot path)',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('data', metavar='DIR', nargs='?', default='synthetic', help='path to dataset')
    p.add_argument('--dataset-format', default='sequential', metavar='STR', help='dataset format')
    p.add_argument('-j', '--workers', default=0, type=int, metavar='N', help='number of data loading workers')
    p.add_argument('--epochs', default=200, type=int, metavar='N', help='number of total epochs to run')
    p.add_argument('--epoch_size', default=0, type=int, metavar='N', help='manual epoch size')
    p.add_argument('--batch_size', default=24, type=int, metavar='N', help='mini-batch size (per GPU)')
    p.add_argument('--lr', default=0.00002, type=float, metavar='LR', help='initial learning rate')
    p.add_argument('--momentum', default=0.9, type=float, metavar='M', help='alpha parameter for adam')
    p.add_argument('--beta', default=0.999, type=float, metavar='M', help='beta parameters for adam')
    p.add_argument('--weight-decay', '--wd', default=0, type=float, metavar='W',
                   help='accepted for compatibility; like the reference the optimiser uses 5e-4')
    p.add_argument('--print-freq', default=10, type=int, metavar='N', help='print frequency')
    p.add_argument('-e', '--evaluate', dest='evaluate', action='store_true', help='evaluate model on validation set')
    p.add_argument('-i', '--img_test', dest='img_test', action='store_true', help='img test on validation set')
    p.add_argument('-r', '--real_test', dest='real_test', action='store_true', help='test on Eigen test split')
    p.add_argument('--seed', default=0, type=int, help='seed for random functions, and network initialization')
    p.add_argument('--log-summary', default='progress_log_summary.csv', metavar='PATH')
    p.add_argument('--log-full', default='progress_log_full.csv', metavar='PATH')
    p.add_argument('--result_dir', type=str, default='./AE_results')
    p.add_argument('--model_dir', type=str, default='./AE_trained_model_lr0000')
    p.add_argument('--RtoD_model_dir', type=str,
                   default='./AE_RtoD_trained_model_lr0004_color_nonMulti/epoch_18_AE_depth_loss_0.2561.pkl')
    p.add_argument('--gpu_num', type=str, default="2")
    p.add_argument('--norm', type=str, default="Batch")
    p.add_argument('--mode', type=str, default="DtoD")
    p.add_argument('--height', type=int, default=128)
    p.add_argument('--width', type=int, default=416)
    p.add_argument('--dataset', type=str, default="KITTI")
    p.add_argument('--img_save', action='store_true', help='result image save')
    # --- additions of this build ---
    p.add_argument('--synthetic', action='store_true', help='train on synthetic KITTI-shaped batches (no dataset)')
    p.add_argument('--local_rank', type=int, default=0, help='set by the launcher; RANK/LOCAL_RANK env win')
    p.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'],
                   help='activation / MFMA operand storage type: fp32 (the reference\'s) or bf16 with fp32 accumulation, '
                        'fp32 master weights, BatchNorm statistics, losses and Adam')
    p.add_argument('--augment', action='store_true',
                   help='with --synthetic: feed synthetic RAW uint8 samples through the GPU augmentation kernel '
                        '(flip / scale-crop / normalise) instead of ready-made tensors')
    p.add_argument('--latent_grad', action='store_true',
                   help='RtoD: let the latent loss back-propagate through the frozen guide into the trained network '
                        '(the guided training of the paper; the reference as shipped computes it under no_grad, value only)')
    p.add_argument('--faithful_guide', action='store_true',
                   help='RtoD: run the frozen guide as two full forwards like the reference (default: one '
                        'batched encoder-only pass, identical features)')
    p.add_argument('--global_berhu', action='store_true',
                   help='all-reduce(MAX) the BerHu threshold like DataParallel\'s gathered batch (SURVEY 8(e))')
    return p


parser = build_parser()


def parse_args(argv=None):
    return parser.parse_args(argv)
