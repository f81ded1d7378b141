"""Drop-in for the reference's src/torch_core/mle.py: maximum-likelihood IRT (point estimates of every person's ability and
every item's parameters in two embeddings, models.py:22-97), masked mean binary cross-entropy (mle.py:192-197).

    python -m vibo_amd.torch_core.mle --irt-model 2pl --dataset 2pl_simulation --num-person 10000 --num-item 100 --cuda

Same flags (mle.py:24-76), out-dir name ``mle_{irt}_{dataset}_{P}person_{I}item_{maxP}maxperson_{maxI}maxitem_{perc}maskperc_
{A}ability_seed{seed}`` (mle.py:104-116) and checkpoint layout (model_state_dict, epoch, args, total_iters + infer_dict
{ability, item_feat}, missing_imputation_accuracy; mle.py:276-333).  The loss runs through the fused ELBO kernel
(`MLE_*PL.nll_step`): nothing of size persons x items is materialised per step.
"""
import argparse
import math
import os
import time

import numpy as np
import torch

from .. import config, ops
from ..datasets import artificially_mask_dataset, load_dataset
from ..utils import save_checkpoint
from . import vibo as _cli
from .models import MLE_1PL, MLE_2PL, MLE_3PL


def build_parser():
    p = argparse.ArgumentParser(description='maximum-likelihood IRT on MI355X (drop-in for src/torch_core/mle.py)')
    p.add_argument('--irt-model', type=str, default='1pl', choices=['1pl', '2pl', '3pl'])
    p.add_argument('--dataset', type=str, default='1pl_simulation',
                   choices=['1pl_simulation', '2pl_simulation', '3pl_simulation', 'critlangacq', 'duolingo', 'wordbank',
                            'pisa2015_science', 'score_matrix'])
    p.add_argument('--ability-dim', type=int, default=1)
    p.add_argument('--no-infer-dict', action='store_true', default=False)
    p.add_argument('--no-test', action='store_true', default=False)
    p.add_argument('--num-person', type=int, default=1000)
    p.add_argument('--num-item', type=int, default=100)
    p.add_argument('--hidden-dim', type=int, default=64)
    p.add_argument('--max-num-person')
    p.add_argument('--max-num-item')
    p.add_argument('--out-dir', type=str, default=config.OUT_DIR)
    p.add_argument('--lr', type=float, default=5e-3)
    p.add_argument('--batch-size', type=int, default=16, metavar='N')
    p.add_argument('--epochs', type=int, default=100, metavar='N')
    p.add_argument('--num-workers', type=int, default=0)
    p.add_argument('--max-iters', type=int, default=-1, metavar='N')
    p.add_argument('--artificial-missing-perc', type=float, default=0.)
    p.add_argument('--seed', type=int, default=42, metavar='S')
    p.add_argument('--gpu-device', type=int, default=0)
    p.add_argument('--cuda', action='store_true', default=False)
    p.add_argument('--row-format', choices=['auto', 'f32', 'codes'], default='auto',
                   help="device-resident rows: fp32 responses + mask bytes, or one byte per cell (see the VIBO CLI)")
    return p


def epoch_loss(model, optimizer, data, batch_size, train, index_of=lambda rows: rows):
    """One pass over a resident split; minibatch loss weighted by its size (AverageMeter semantics, mle.py:200,224)."""
    model.train(train)
    wsum, count = torch.zeros((), device=data.device), 0
    for rows in data.batches(batch_size, shuffle=train):
        if train:
            optimizer.zero_grad(set_to_none=True)
            loss = model.nll_step(index_of(rows), data.response, data.mask, row_index=rows)
            loss.backward()
            optimizer.step()
        else:
            with torch.no_grad():
                loss = model.nll_step(index_of(rows), data.response, data.mask, row_index=rows)
        wsum += loss.detach() * rows.numel()
        count += rows.numel()
    return float(wsum) / max(1, count)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.artificial_missing_perc:
        args.no_infer_dict = False                                           # mle.py:78-79
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    if config.IS_REAL_WORLD[args.dataset]:                                   # mle.py:84-98
        args.num_person = args.num_item = None
        args.max_num_person = int(args.max_num_person) if args.max_num_person is not None else None
        args.max_num_item = int(args.max_num_item) if args.max_num_item is not None else None
    else:
        args.max_num_person = args.max_num_item = None
    args.out_dir = os.path.join(args.out_dir, 'mle_{}_{}_{}person_{}item_{}maxperson_{}maxitem_{}maskperc_{}ability_seed{}'.format(
        args.irt_model, args.dataset, args.num_person, args.num_item, args.max_num_person, args.max_num_item,
        args.artificial_missing_perc, args.ability_dim, args.seed))
    os.makedirs(args.out_dir, exist_ok=True)
    device = torch.device('cuda', args.gpu_device) if args.cuda else torch.device('cpu')
    if args.cuda:
        torch.cuda.set_device(args.gpu_device)
    kw = dict(num_person=args.num_person, num_item=args.num_item, ability_dim=args.ability_dim,
              max_num_person=args.max_num_person, max_num_item=args.max_num_item)
    train_dataset = load_dataset(args.dataset, train=True, **kw)
    test_dataset = load_dataset(args.dataset, train=False, **kw)
    if args.artificial_missing_perc > 0:
        train_dataset = artificially_mask_dataset(train_dataset, args.artificial_missing_perc)
    num_person, num_item = train_dataset.num_person, train_dataset.num_item
    fmt = args.row_format
    codes_ok = bool(args.cuda) and 4 <= num_item <= 32767
    fmt = ('codes' if codes_ok else 'f32') if fmt == 'auto' else fmt
    train = _cli.ResidentSplit(train_dataset, device, None, fmt)
    test = _cli.ResidentSplit(test_dataset, device, None, fmt)
    n_batches = train.num_batches(args.batch_size)
    if args.max_iters != -1:
        args.epochs = int(math.ceil(args.max_iters / float(n_batches)))
        print(f'Found MAX_ITERS={args.max_iters}, setting EPOCHS={args.epochs}')
    model = {'1pl': MLE_1PL, '2pl': MLE_2PL, '3pl': MLE_3PL}[args.irt_model](args.ability_dim, num_person, num_item).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr)

    best_loss = np.inf
    train_losses, test_losses, train_times = np.zeros(args.epochs), np.zeros(args.epochs), np.zeros(args.epochs)
    for epoch in range(args.epochs):
        t0 = time.time()
        train_losses[epoch] = epoch_loss(model, optimizer, train, args.batch_size, True)
        if args.cuda:
            torch.cuda.synchronize()
        train_times[epoch] = t0 - time.time()                                # negative, like the reference (mle.py:268)
        print('====> Train Epoch: {} Loss: {:.4f}'.format(epoch, train_losses[epoch]))
        if not args.no_test:
            # the reference scores the held-out persons against the ability rows of the same index (mle.py:210-232)
            test_losses[epoch] = epoch_loss(model, None, test, args.batch_size, False, index_of=lambda rows: rows % num_person)
            print('====> Test Epoch: {} Loss: {:.4f}'.format(epoch, test_losses[epoch]))
            score = test_losses[epoch]
        else:
            score = train_losses[epoch]
        is_best = score < best_loss
        best_loss = min(score, best_loss)
        save_checkpoint({'model_state_dict': model.state_dict(), 'epoch': epoch, 'args': args,
                         'total_iters': n_batches * args.epochs}, is_best, folder=args.out_dir)
        np.save(os.path.join(args.out_dir, 'train_losses.npy'), train_losses)
        np.save(os.path.join(args.out_dir, 'train_times.npy'), train_times)
        if not args.no_test:
            np.save(os.path.join(args.out_dir, 'test_losses.npy'), test_losses)

    for name in ('checkpoint.pth.tar', 'model_best.pth.tar'):
        path = os.path.join(args.out_dir, name)
        if not os.path.exists(path):
            continue
        ckpt = torch.load(path, weights_only=False)
        model.load_state_dict(ckpt['model_state_dict'])
        if not args.no_infer_dict:
            with torch.no_grad():
                ability, item_feat = model.ability.weight.detach().cpu(), model.item_feat.weight.detach().cpu()
                ckpt['infer_dict'] = {'ability': ability, 'item_feat': [item_feat] * n_batches}      # mle.py:235-259: one (identical) entry per batch
                if args.artificial_missing_perc > 0:
                    inferred = model.decode(model.ability.weight, model.item_feat.weight).squeeze(2).cpu()
                    ckpt['missing_imputation_accuracy'] = _cli.imputation_accuracy(
                        inferred, train_dataset.missing_indices, train_dataset.missing_labels)
        torch.save(ckpt, path)
    print(f'Saved to {args.out_dir}')
    return args.out_dir


if __name__ == '__main__':
    main()
