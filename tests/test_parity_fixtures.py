"""Fixtures generated from the REAL reference (tools/gen_golden.py, build container) that pin host-side behaviour the
kernels do not touch: constructor / weights_init draw order, the CritLangAcq loader, and the oracle's 3PL arithmetic in
the Bernoulli clamp band."""
import io
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from vibo_amd import config, datasets
from vibo_amd.torch_core.models import VIBO_1PL, VIBO_2PL, VIBO_3PL


def test_seeded_construction_draws_the_reference_numbers():
    """torch.manual_seed(s); VIBO_*PL(...) gives the reference's state_dict bitwise and leaves the generator in the same
    state (models.py:281-329 constructor order, 512-518 weights_init; embeddings N(0,1), flows randn)."""
    z = np.load(os.path.join(GOLDEN_DIR, 'seeded_init.npz'))
    for m in json.loads(str(z['meta'])):
        torch.manual_seed(m['seed'])
        cls = {1: VIBO_1PL, 2: VIBO_2PL, 3: VIBO_3PL}[m['irt_model']]
        model = cls(m['ability_dim'], m['num_item'], ability_merge=m['ability_merge'],
                    conditional_posterior=m['conditional_posterior'], n_norm_flows=m['n_norm_flows'])
        sd = model.state_dict()
        ref_keys = [k[len(m['name']) + 1:] for k in z.files if k.startswith(m['name'] + '.') and not k.endswith('next_randn')]
        assert sorted(sd.keys()) == sorted(ref_keys), m['name']
        for k in ref_keys:
            assert np.array_equal(sd[k].numpy(), z[f"{m['name']}.{k}"]), (m['name'], k)
        assert np.array_equal(torch.randn(4).numpy(), z[f"{m['name']}.next_randn"]), m['name']


def test_critlangacq_loader_matches_the_reference(tmp_path, monkeypatch):
    """datasets.py:283-440 on a synthetic data.csv (95 q* columns selected by name from a scrambled file with a decoy
    column, RandomState(42) row shuffle, 80/20 split, -1 = missing, max_num_person / max_num_item caps)."""
    import pandas as pd
    z = np.load(os.path.join(GOLDEN_DIR, 'critlangacq_loader.npz'))
    cols = json.loads(str(z['columns']))
    df = pd.DataFrame(z['table'], columns=cols)
    os.makedirs(tmp_path / 'critlangacq')
    df.to_csv(tmp_path / 'critlangacq' / 'data.csv', index=False)
    monkeypatch.setattr(config, 'DATA_DIR', str(tmp_path))
    for name, kw in (('train', dict(train=True)), ('test', dict(train=False)),
                     ('train_cap', dict(train=True, max_num_person=100, max_num_item=40))):
        d = datasets.load_dataset('critlangacq', num_person=None, num_item=None, ability_dim=1,
                                  max_num_person=kw.get('max_num_person'), max_num_item=kw.get('max_num_item'), train=kw['train'])
        assert np.array_equal(np.asarray(d.response).reshape(z[f'{name}.response'].shape), z[f'{name}.response']), name
        assert np.array_equal(np.asarray(d.mask).reshape(z[f'{name}.mask'].shape) != 0, z[f'{name}.mask'] != 0), name
        assert np.array_equal(np.asarray(d.item_id), z[f'{name}.item_id']), name
        assert (d.num_person, d.num_item) == z[f'{name}.response'].shape
        idx, r, iid, m = d[3]
        assert np.array_equal(r.numpy().reshape(-1), z[f'{name}.getitem3.response'].reshape(-1))
        assert np.array_equal(iid.numpy().reshape(-1), z[f'{name}.getitem3.item_id'].reshape(-1))
        assert np.array_equal(m.numpy().reshape(-1), z[f'{name}.getitem3.mask'].reshape(-1))
        r2, m2 = d.matrix()
        assert r2.shape == z[f'{name}.response'].shape and m2.dtype == np.bool_


def test_oracle_3pl_in_the_clamp_band_is_the_reference():
    """oracle.irt_link + masked_bernoulli_ll on the 3PL saturation grid: p, log-lik and both gradients as the reference
    computes them (utils.py:46-49 on p = g + (1 - g) sigmoid(l), models.py:748-766), bit for bit."""
    from oracle import vibo_oracle as O
    z = np.load(os.path.join(GOLDEN_DIR, 'saturation_3pl.npz'))
    l = torch.from_numpy(z['logit'])
    for gi, gl in enumerate(z['guess_logit']):
        for x in (0, 1):
            item = torch.stack([torch.ones_like(l), l, torch.full_like(l, float(gl))], dim=1).requires_grad_(True)
            p = O.irt_link(3, torch.zeros(1, 1), item)
            ll = O.masked_bernoulli_ll(torch.full_like(p, float(x)), torch.ones_like(p), p)
            g, = torch.autograd.grad(ll.sum(), item)
            assert np.array_equal(p.detach().reshape(-1).numpy(), z[f'p_g{gi}'])
            assert np.array_equal(ll.detach().reshape(-1).numpy(), z[f'll_g{gi}_x{x}'])
            assert np.array_equal(g[:, 1].numpy(), z[f'dll_db_g{gi}_x{x}'])
            assert np.array_equal(g[:, 2].numpy(), z[f'dll_dguess_g{gi}_x{x}'])
