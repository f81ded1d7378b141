"""Synthetic IRT datasets in the reference's on-disk format.

Reference: src/simulate.py:40-58 saves {'response','ability','item_feat'} to
DATA_DIR/{irt}_simulation_{P}person_{I}item_{A}ability/simulation.pth, generated
by the Pyro models of src/pyro_core/models.py:25-159: ability ~ N(0,1) [P,A],
item_feat ~ N(0,1) [I,D], response ~ Bernoulli(link) [P,I,1].  Pyro is not a
dependency here: the same draws are made with torch in the same order.

    python -m vibo_amd.simulate --irt-model 2pl --num-person 10000 --num-item 100
"""
import argparse
import os

import torch

from .config import DATA_DIR
from .ops import item_feat_dim


def simulation_dir(irt_model, num_person, num_item, ability_dim, nonlinear=False, data_dir=None):
    d = os.path.join(data_dir or DATA_DIR, f'{irt_model}_simulation_{num_person}person_{num_item}item_{ability_dim}ability')
    return d + '_nonlinear' if nonlinear else d


def link_probability(irt_num, ability, item_feat, nonlinear=False):
    A = ability.shape[1]
    if irt_num == 1:
        logit = ability.sum(1, keepdim=True) + item_feat.t()
    else:
        logit = ability @ (-item_feat[:, :A].t()) + item_feat[:, A:A + 1].t()
    if nonlinear:
        logit = logit.pow(2)
    p = torch.sigmoid(logit)
    if irt_num == 3:
        guess = torch.sigmoid(item_feat[:, A + 1:A + 2]).t()
        p = guess + (1.0 - guess) * p
    return p


def generate(irt_model, num_person, num_item, ability_dim, seed=42, nonlinear=False, chunk=65536):
    irt_num = int(str(irt_model)[0])
    g = torch.Generator().manual_seed(seed)
    ability = torch.randn(num_person, ability_dim, generator=g)
    item_feat = torch.randn(num_item, item_feat_dim(irt_num, ability_dim), generator=g)
    response = torch.empty(num_person, num_item, 1)
    for s in range(0, num_person, chunk):
        p = link_probability(irt_num, ability[s:s + chunk], item_feat, nonlinear)
        response[s:s + chunk, :, 0] = torch.bernoulli(p, generator=g)
    return {'response': response, 'ability': ability, 'item_feat': item_feat}


def synthetic_critlangacq_csv(path, num_person=2500, seed=7, native_missing=0.03):
    """A stand-in for DATA_DIR/critlangacq/data.csv (the real file, datasets.py:283-440, is not redistributable): the 95
    item columns of the loader -- in a scrambled file order, with a decoy q-column and metadata columns in between -- filled
    with 2PL responses (ability_dim 1, `generate`'s draws) and `native_missing` of the cells set to -1.  Deterministic in
    (num_person, seed): tools/gen_cli_golden.py feeds the same file to the reference CLI."""
    import numpy as np
    import pandas as pd
    from .datasets import critlangacq_item_keys
    keys = list(critlangacq_item_keys())
    sim = generate('2pl', num_person, len(keys), 1, seed=seed)
    resp = sim['response'][:, :, 0].numpy().astype(np.int64)
    rs = np.random.RandomState(seed)
    resp[rs.rand(*resp.shape) < native_missing] = -1
    cols = {'id': np.arange(num_person), 'age': rs.randint(7, 80, num_person), 'education': rs.randint(0, 6, num_person)}
    order = list(range(len(keys)))
    rs.shuffle(order)
    for pos, k in enumerate(order):
        if pos == 7:
            cols['q4_decoy'] = (rs.rand(num_person) < 0.5).astype(np.int64)      # a q-column that is NOT one of the 95 items
        cols[keys[k]] = resp[:, k]
    os.makedirs(os.path.dirname(path), exist_ok=True)
    pd.DataFrame(cols).to_csv(path, index=False)
    return path


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--irt-model', type=str, default='3pl', choices=['1pl', '2pl', '3pl'])
    ap.add_argument('--num-person', type=int, default=1000)
    ap.add_argument('--num-item', type=int, default=100)
    ap.add_argument('--ability-dim', type=int, default=1)
    ap.add_argument('--nonlinear', action='store_true', default=False)
    ap.add_argument('--seed', type=int, default=42)
    args = ap.parse_args(argv)
    out = simulation_dir(args.irt_model, args.num_person, args.num_item, args.ability_dim, args.nonlinear)
    os.makedirs(out, exist_ok=True)
    ds = generate(args.irt_model, args.num_person, args.num_item, args.ability_dim, args.seed, args.nonlinear)
    print(f'Saving to {out}')
    torch.save(ds, os.path.join(out, 'simulation.pth'))


if __name__ == '__main__':
    main()
