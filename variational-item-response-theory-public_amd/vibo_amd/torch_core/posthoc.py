"""Post-hoc enrichment of a saved VIBO checkpoint -- the drop-ins for the reference's standalone scripts
src/torch_core/infer.py (infer_dict), marginal.py (train_logp / test_logp) and predictives.py (posterior_predict_samples,
missing_imputation_accuracy).  Like them: the model, dataset and device are rebuilt from the checkpoint's own `args`
Namespace, the result is written back into the checkpoint file.  The split stays resident on the device and the
model.encode / log_marginal / decode calls of those scripts (infer.py:74-75, marginal.py:81-85, predictives.py:87-103) run
through the fused kernels (vibo_encode, vibo_elbo_multi_forward, vibo_decode_mean).

    python -m vibo_amd.torch_core.infer        out/<run>/checkpoint.pth.tar
    python -m vibo_amd.torch_core.marginal     out/<run>/checkpoint.pth.tar
    python -m vibo_amd.torch_core.predictives  out/<run>/checkpoint.pth.tar [--num-posterior-samples 200]
"""
import argparse

import torch

from ..datasets import artificially_mask_dataset, load_dataset
from . import vibo as _cli
from .models import VIBO_1PL, VIBO_2PL, VIBO_3PL


def _dataset(args, train):
    name = args.dataset if getattr(args, 'response_dist', 'bernoulli') == 'bernoulli' else f'{args.dataset}_continuous'
    return load_dataset(name, train=train, num_person=args.num_person, num_item=args.num_item, ability_dim=args.ability_dim,
                        max_num_person=args.max_num_person, max_num_item=args.max_num_item)


def _model(args, num_item, state_dict, device):
    model = {'1pl': VIBO_1PL, '2pl': VIBO_2PL, '3pl': VIBO_3PL}[args.irt_model](
        args.ability_dim, num_item, hidden_dim=args.hidden_dim, ability_merge=args.ability_merge,
        conditional_posterior=args.conditional_posterior, generative_model=args.generative_model,
        response_dist=getattr(args, 'response_dist', 'bernoulli'),
        replace_missing_with_prior=not getattr(args, 'drop_missing', False),
        n_norm_flows=getattr(args, 'n_norm_flows', 0)).to(device)
    model.load_state_dict(state_dict)
    return model


def _split(args, dataset, device):
    num_item = dataset.num_item
    fmt = getattr(args, 'row_format', 'auto')
    ok = bool(args.cuda) and 4 <= num_item <= 32767 and getattr(args, 'ability_dim', 1) <= 8
    fmt = ('codes' if ok else 'f32') if fmt == 'auto' else fmt
    return _cli.ResidentSplit(dataset, device, None, fmt)


def run(what, argv=None):
    p = argparse.ArgumentParser(description=f'{what} for a saved VIBO checkpoint (written back into the file)')
    p.add_argument('checkpoint', type=str)
    if what == 'predictives':
        p.add_argument('--num-posterior-samples', type=int, default=200)            # predictives.py:23
    cli_args = p.parse_args(argv)
    ckpt = torch.load(cli_args.checkpoint, weights_only=False)
    args = ckpt['args']
    if what == 'predictives':
        args.num_posterior_samples = cli_args.num_posterior_samples
    device = torch.device('cuda', args.gpu_device) if args.cuda else torch.device('cpu')
    if args.cuda:
        torch.cuda.set_device(args.gpu_device)
    train_dataset = _dataset(args, True)
    if what == 'predictives' and args.artificial_missing_perc > 0:
        train_dataset = artificially_mask_dataset(train_dataset, args.artificial_missing_perc)     # predictives.py:56-60
    model = _model(args, train_dataset.num_item, ckpt['model_state_dict'], device)
    train = _split(args, train_dataset, device)
    if what == 'infer':                                                               # infer.py:59-107
        ckpt['infer_dict'] = _cli.infer_dict(model, train, args.batch_size)
    elif what == 'marginal':                                                          # marginal.py:70-117
        test = _split(args, _dataset(args, False), device)
        ckpt['train_logp'] = _cli.log_marginal_density(model, train, args, args.batch_size)
        ckpt['test_logp'] = _cli.log_marginal_density(model, test, args, args.batch_size)
    else:                                                                             # predictives.py:73-147
        samples = _cli.posterior_predictive(model, train, args, args.batch_size, getattr(args, 'store_predictive_samples', False))
        ckpt['posterior_predict_samples'] = samples
        if args.artificial_missing_perc > 0:
            acc = _cli.imputation_accuracy(samples['response'].mean(0).squeeze(-1), train_dataset.missing_indices,
                                           train_dataset.missing_labels)
            ckpt['missing_imputation_accuracy'] = acc
            print(acc)
    torch.save(ckpt, cli_args.checkpoint)
    return ckpt
