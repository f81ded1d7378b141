"""Drop-in for the reference's src/torch_core/vi.py: un-amortized variational inference for IRT (one Gaussian posterior
per person in two embeddings, models.py:100-243) trained on the train split with the same fused ELBO kernel, which takes
the looked-up (mu, logvar) rows as a caller-supplied posterior.

    python -m vibo_amd.torch_core.vi --irt-model 2pl --dataset 2pl_simulation --num-person 10000 --num-item 100 --cuda

Same flags (vi.py:20-78), out-dir name ``vi_{irt}_{dataset}_{P}person_{I}item_{perc}maskperc_{A}ability`` (vi.py:87-96)
and checkpoint layout (vi.py:318-368: model_state_dict, epoch, args + infer_dict, posterior_predict_samples,
missing_imputation_accuracy, train_logp), ``train_losses.npy`` / ``train_times.npy``.  The split stays resident on the
device (one byte per cell by default) and minibatches are row-index vectors, as in the VIBO CLI next door.
(--prior-scale / --reduce-lr / --patience are accepted and, as in the reference script, not used.)
"""
import argparse
import math
import os
import time

import numpy as np
import torch

from .. import config, ops
from ..datasets import artificially_mask_dataset, load_dataset
from ..utils import AverageMeter, save_checkpoint
from . import vibo as _cli
from .models import VI_1PL, VI_2PL, VI_3PL


def build_parser():
    p = argparse.ArgumentParser(description='un-amortized VI for IRT on MI355X (drop-in for src/torch_core/vi.py)')
    p.add_argument('--irt-model', type=str, default='1pl', choices=['1pl', '2pl', '3pl'])
    p.add_argument('--dataset', type=str, default='1pl_simulation',
                   choices=['1pl_simulation', '2pl_simulation', '3pl_simulation', 'critlangacq', 'duolingo', 'wordbank',
                            'pisa2015_science', 'score_matrix'])
    p.add_argument('--ability-dim', type=int, default=1)
    p.add_argument('--artificial-missing-perc', type=float, default=0.)
    p.add_argument('--num-person', type=int, default=1000)
    p.add_argument('--num-item', type=int, default=100)
    p.add_argument('--num-posterior-samples', type=int, default=400)
    p.add_argument('--max-num-person')
    p.add_argument('--max-num-item')
    p.add_argument('--out-dir', type=str, default=config.OUT_DIR)
    p.add_argument('--lr', type=float, default=5e-3)
    p.add_argument('--batch-size', type=int, default=16, metavar='N')
    p.add_argument('--epochs', type=int, default=100, metavar='N')
    p.add_argument('--max-iters', type=int, default=-1, metavar='N')
    p.add_argument('--anneal-kl', action='store_true', default=False)
    p.add_argument('--beta-kl', type=float, default=1.0)
    p.add_argument('--prior-scale', type=float, default=1.)
    p.add_argument('--no-marginal', action='store_true', default=False)
    p.add_argument('--no-predictive', action='store_true', default=False)
    p.add_argument('--reduce-lr', action='store_true', default=False)
    p.add_argument('--patience', type=int, default=100)
    p.add_argument('--seed', type=int, default=42, metavar='S')
    p.add_argument('--gpu-device', type=int, default=0)
    p.add_argument('--cuda', action='store_true', default=False)
    # additions of this implementation
    p.add_argument('--row-format', choices=['auto', 'f32', 'codes'], default='auto',
                   help="device-resident rows: fp32 responses + mask bytes, or one byte per cell (see the VIBO CLI)")
    p.add_argument('--store-predictive-samples', action='store_true', default=False,
                   help='keep all S posterior-predictive samples [S,P,I,1] like the reference (default: their mean)')
    return p


class ResidentVI:
    """The VI model behind the method surface the VIBO CLI's epoch / inference helpers use: person index = row of the
    resident train split."""

    def __init__(self, model):
        self.model, self.spec = model, model.spec

    def train(self):
        self.model.train()

    def eval(self):
        self.model.eval()

    def elbo_step(self, response, mask, annealing_factor=1.0, row_index=None):
        outs = self.model(row_index, response, mask, row_index=row_index)
        return self.model.elbo(*outs, annealing_factor=annealing_factor)        # vi.py:172-173

    def encode(self, response, mask, row_index=None):
        return self.model.encode(row_index)

    def decode(self, ability, item_feat):
        return self.model.decode(ability, item_feat)


def log_marginal_density(model, data, args, batch_size):
    """vi.py:187-215: batch-level importance-weighted bound, batch-size weighted average."""
    meter = AverageMeter()
    model.eval()
    with torch.no_grad():
        for rows in data.batches(batch_size, shuffle=False):
            log_w = torch.stack([-model.elbo(*model(rows, data.response, data.mask, row_index=rows), annealing_factor=1,
                                             use_kl_divergence=False) for _ in range(args.num_posterior_samples)])
            meter.update(float(torch.logsumexp(log_w, 0) - math.log(args.num_posterior_samples)), rows.numel())
    print('====> Marginal: {:.4f}'.format(meter.avg))
    return meter.avg


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.artificial_missing_perc > 0:
        args.no_predictive = False                                           # vi.py:80-81
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    args.out_dir = os.path.join(args.out_dir, 'vi_{}_{}_{}person_{}item_{}maskperc_{}ability'.format(
        args.irt_model, args.dataset, args.num_person, args.num_item, args.artificial_missing_perc, args.ability_dim))
    os.makedirs(args.out_dir, exist_ok=True)
    device = torch.device('cuda', args.gpu_device) if args.cuda else torch.device('cpu')
    if args.cuda:
        torch.cuda.set_device(args.gpu_device)
    train_dataset = load_dataset(args.dataset, train=True, num_person=args.num_person, num_item=args.num_item,
                                 ability_dim=args.ability_dim, max_num_person=args.max_num_person,
                                 max_num_item=args.max_num_item)
    if args.artificial_missing_perc > 0:
        train_dataset = artificially_mask_dataset(train_dataset, args.artificial_missing_perc)
    num_person, num_item = train_dataset.num_person, train_dataset.num_item
    row_format = args.row_format
    codes_ok = bool(args.cuda) and 4 <= num_item <= 32767
    if row_format == 'auto':
        row_format = 'codes' if codes_ok else 'f32'
    elif row_format == 'codes' and not codes_ok:
        raise SystemExit('--row-format codes needs --cuda and 4..32767 items')
    train = _cli.ResidentSplit(train_dataset, device, None, row_format)
    n_batches = train.num_batches(args.batch_size)
    if args.max_iters != -1:
        args.epochs = int(math.ceil(args.max_iters / float(n_batches)))
        print(f'Found MAX_ITERS={args.max_iters}, setting EPOCHS={args.epochs}')
    model = {'1pl': VI_1PL, '2pl': VI_2PL, '3pl': VI_3PL}[args.irt_model](args.ability_dim, num_person, num_item).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr)
    face = ResidentVI(model)

    best_loss = np.inf
    train_losses, train_times = np.zeros(args.epochs), np.zeros(args.epochs)
    for epoch in range(args.epochs):
        t0 = time.time()
        train_loss = _cli.train_epoch(face, optimizer, train, args, epoch, args.batch_size)
        if args.cuda:
            torch.cuda.synchronize()
        train_losses[epoch] = train_loss
        train_times[epoch] = t0 - time.time()                                # negative, like the reference (vi.py:312)
        is_best = train_loss < best_loss
        best_loss = min(train_loss, best_loss)
        save_checkpoint({'model_state_dict': model.state_dict(), 'epoch': epoch, 'args': args}, is_best, folder=args.out_dir)
        np.save(os.path.join(args.out_dir, 'train_losses.npy'), train_losses)
        np.save(os.path.join(args.out_dir, 'train_times.npy'), train_times)

    for name in ('checkpoint.pth.tar', 'model_best.pth.tar'):
        path = os.path.join(args.out_dir, name)
        if not os.path.exists(path):
            continue
        ckpt = torch.load(path, weights_only=False)
        model.load_state_dict(ckpt['model_state_dict'])
        infer = _cli.infer_dict(face, train, args.batch_size)
        for k in ('item_feat_mu', 'item_feat_logvar'):      # vi.py:289-290 stacks the (identical) per-batch item tensors:
            infer[k] = infer[k].cpu().unsqueeze(0).expand(n_batches, -1, -1)      # [n_batches, I, D] (stride 0: stored once)
        if not args.no_predictive:
            samples = _cli.posterior_predictive(face, train, args, args.batch_size, args.store_predictive_samples)
            ckpt['posterior_predict_samples'] = samples
            if args.artificial_missing_perc > 0:
                acc = _cli.imputation_accuracy(samples['response'].mean(0).squeeze(-1), train_dataset.missing_indices,
                                               train_dataset.missing_labels)
                ckpt['missing_imputation_accuracy'] = acc
                print(f'Missing Imputation Accuracy from samples: {acc}')
        if not args.no_marginal:
            ckpt['train_logp'] = log_marginal_density(model, train, args, args.batch_size)
        ckpt['infer_dict'] = infer
        torch.save(ckpt, path)
    print(f'Saved to {args.out_dir}')
    return args.out_dir


if __name__ == '__main__':
    main()
