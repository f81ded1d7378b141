import glob
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
PKG_DIR = os.path.join(ROOT, 'variational-item-response-theory-public_amd')
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden_case_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, 'case_*.npz')))


class Golden:
    """One tests/golden/case_*.npz fixture (generated from the reference by
    tools/gen_golden.py), as torch tensors."""

    def __init__(self, path):
        z = np.load(path)
        self.path = path
        self.meta = json.loads(str(z['meta']))
        self.response = torch.from_numpy(z['response'].astype(np.float32))
        self.mask = torch.from_numpy(z['mask'].astype(np.uint8))
        self.eps_item = torch.from_numpy(z['eps_item'])
        self.eps_ability = torch.from_numpy(z['eps_ability'])
        self.sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd.')}
        self.out = {k[4:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith('out.')}
        self.grad = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('grad.')}
        self.adam1 = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('adam1.')}
        self.adam3 = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('adam3.')}

    @property
    def cfg(self):
        m = self.meta
        return dict(irt_model=m['irt_model'], ability_dim=m['ability_dim'],
                    conditional_posterior=m['conditional_posterior'],
                    replace_missing_with_prior=m['replace_missing_with_prior'],
                    n_norm_flows=m['n_norm_flows'],
                    annealing_factor=m['annealing_factor'],
                    use_kl_divergence=m['use_kl_divergence'],
                    **({'generative_model': m['generative_model']} if m.get('generative_model', 'irt') != 'irt' else {}))

    def __repr__(self):
        return os.path.basename(self.path)


@pytest.fixture(params=golden_case_files(), ids=lambda p: os.path.basename(p)[5:-4])
def golden(request):
    return Golden(request.param)


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
