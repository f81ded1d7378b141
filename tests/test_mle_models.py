"""Maximum-likelihood siblings (reference models.py:22-97, mle.py): MLE_1PL/2PL/3PL.  Goldens (tests/golden/mle_*.npz) come
from the real reference classes and the loss of mle.py:192-197."""
import glob
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN_DIR, rel_err
from oracle import cpu_backend
from vibo_amd import config, ops
from vibo_amd.torch_core.models import MLE_1PL, MLE_2PL, MLE_3PL

CLS = {1: MLE_1PL, 2: MLE_2PL, 3: MLE_3PL}
FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, 'mle_*.npz')))


@pytest.fixture(params=FILES, ids=lambda p: os.path.basename(p)[:-4])
def mle_golden(request):
    z = np.load(request.param)
    g = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files if k != 'meta'}
    g['meta'] = json.loads(str(z['meta']))
    g['response'] = g['response'].float()
    return g


def test_golden_files_exist():
    assert len(FILES) == 3


def run_model(g, device):
    m = g['meta']
    model = CLS[m['irt_model']](m['ability_dim'], m['num_person'], m['num_item'])
    sd = {k[3:]: v for k, v in g.items() if k.startswith('sd.')}
    assert list(model.state_dict().keys()) == list(sd.keys()) == ['ability.weight', 'item_feat.weight']
    model.load_state_dict(sd, strict=True)
    model = model.to(device)
    idx = g['index'].to(device)
    resp, mask = g['response'].to(device), g['mask'].bool().to(device)
    # (1) the reference's own call pattern: materialised response_mu, masked BCE outside the model (mle.py:193-197)
    response_mu = model(idx, resp.unsqueeze(2), mask.long().unsqueeze(2))
    assert (response_mu.detach().squeeze(2).cpu() - g['out.response_mu']).abs().max() < 2e-6
    loss = (F.binary_cross_entropy(response_mu, resp.clamp(min=0).unsqueeze(2), reduction='none') * mask.unsqueeze(2)).mean()
    assert rel_err(loss.detach().cpu(), g['out.loss']) < 1e-5
    loss.backward()
    for name, p in model.named_parameters():
        assert rel_err(p.grad.cpu(), g['grad.' + name]) < 1e-4, name
    # (2) the same loss through the fused kernel
    model.zero_grad()
    fast = model.nll_step(idx, resp, mask)
    assert rel_err(fast.detach().cpu(), g['out.loss']) < 2e-5
    fast.backward()
    for name, p in model.named_parameters():
        assert rel_err(p.grad.cpu(), g['grad.' + name]) < 3e-4, name
    assert (model.decode(*model.encode(idx)).squeeze(2).cpu() - g['out.response_mu']).abs().max() < 2e-6
    return model


def test_models_match_reference_on_the_cpu_stand_in(mle_golden):
    restore = cpu_backend.install(ops)
    try:
        run_model(mle_golden, torch.device('cpu'))
    finally:
        restore()


@pytest.mark.gpu
def test_models_match_reference_on_the_gpu(mle_golden):
    model = run_model(mle_golden, torch.device('cuda:0'))
    g, dev = mle_golden, torch.device('cuda:0')
    idx = g['index'].to(dev)
    P = g['meta']['num_person']
    full_r = torch.zeros(P, g['response'].shape[1], device=dev)
    full_m = torch.zeros(P, g['response'].shape[1], dtype=torch.bool, device=dev)
    full_r[idx], full_m[idx] = g['response'].to(dev), g['mask'].bool().to(dev)
    a = model.nll_step(idx, g['response'].to(dev), g['mask'].bool().to(dev))
    b = model.nll_step(idx, ops.pack_cell_codes(full_r, full_m), None, row_index=idx)      # resident cell codes + gather
    assert rel_err(b.detach().cpu(), a.detach().cpu()) < 1e-6


def _run_cli(tmp_path, monkeypatch, cuda):
    from vibo_amd.torch_core import mle
    monkeypatch.setattr(config, 'DATA_DIR', str(tmp_path / 'data'))
    monkeypatch.setattr(config, 'OUT_DIR', str(tmp_path / 'out'))
    argv = ['--irt-model', '2pl', '--dataset', '2pl_simulation', '--num-person', '400', '--num-item', '20', '--epochs', '6',
            '--batch-size', '16', '--lr', '0.05', '--artificial-missing-perc', '0.2', '--out-dir', str(tmp_path / 'out')]
    out_dir = mle.main(argv + (['--cuda'] if cuda else []))
    assert os.path.basename(out_dir) == 'mle_2pl_2pl_simulation_400person_20item_Nonemaxperson_Nonemaxitem_0.2maskperc_1ability_seed42'
    ck = torch.load(os.path.join(out_dir, 'checkpoint.pth.tar'), weights_only=False)
    assert {'model_state_dict', 'epoch', 'args', 'total_iters', 'infer_dict', 'missing_imputation_accuracy'} <= set(ck)
    assert ck['infer_dict']['ability'].shape == (320, 1) and ck['infer_dict']['item_feat'][0].shape == (20, 2)
    assert 0.5 < ck['missing_imputation_accuracy'] <= 1.0
    tr, te = np.load(os.path.join(out_dir, 'train_losses.npy')), np.load(os.path.join(out_dir, 'test_losses.npy'))
    assert np.isfinite(tr).all() and np.isfinite(te).all() and tr[-1] < tr[0]


def test_mle_cli_on_the_cpu_stand_in(tmp_path, monkeypatch):
    restore = cpu_backend.install(ops)
    try:
        _run_cli(tmp_path, monkeypatch, cuda=False)
    finally:
        restore()


@pytest.mark.gpu
def test_mle_cli_on_the_gpu(tmp_path, monkeypatch):
    _run_cli(tmp_path, monkeypatch, cuda=True)


@pytest.mark.gpu
def test_trained_mle_model_matches_the_reference_mle_run(tmp_path, monkeypatch):
    """The reference's mle.py run end to end on CPU (tools/gen_cli_golden.py mle -> tests/golden/cli_trained_mle_2pl.npz; complete
    data: its BCE call rejects the -1 targets of hidden cells on current PyTorch) against this mle.py on the GPU on the same seeded
    data.  No sampling noise here, only the minibatch order differs."""
    from vibo_amd import simulate
    from vibo_amd.torch_core import mle
    z = np.load(os.path.join(GOLDEN_DIR, 'cli_trained_mle_2pl.npz'))
    a = json.loads(str(z['meta']))
    monkeypatch.setattr(config, 'DATA_DIR', str(tmp_path / 'data'))
    monkeypatch.setattr(config, 'OUT_DIR', str(tmp_path / 'out'))
    d = simulate.simulation_dir(a['irt'], a['num_person'], a['num_item'], a['ability_dim'], data_dir=str(tmp_path / 'data'))
    os.makedirs(d, exist_ok=True)
    torch.save(simulate.generate(a['irt'], a['num_person'], a['num_item'], a['ability_dim'], seed=a['seed']),
               os.path.join(d, 'simulation.pth'))
    out_dir = mle.main(['--irt-model', a['irt'], '--dataset', f"{a['irt']}_simulation", '--num-person', str(a['num_person']),
                        '--num-item', str(a['num_item']), '--ability-dim', str(a['ability_dim']), '--epochs', str(a['epochs']),
                        '--batch-size', str(a['batch']), '--seed', str(a['seed']), '--lr', str(a['lr']), '--cuda',
                        '--out-dir', str(tmp_path / 'out')])
    assert os.path.basename(out_dir) == a['run_dir']
    tr, te = np.load(os.path.join(out_dir, 'train_losses.npy')), np.load(os.path.join(out_dir, 'test_losses.npy'))
    assert np.abs(tr - z['train_losses']).max() < 0.01 * z['train_losses'].max(), (tr, z['train_losses'])
    assert np.abs(te - z['test_losses']).max() < 0.03 * z['test_losses'].max(), (te, z['test_losses'])
    ck = torch.load(os.path.join(out_dir, 'checkpoint.pth.tar'), weights_only=False)
    assert len(ck['infer_dict']['item_feat']) == int(z['n_item_feat_copies'])
    r = np.corrcoef(ck['infer_dict']['ability'].numpy().ravel(), z['ability'].ravel())[0, 1]
    ri = np.corrcoef(ck['infer_dict']['item_feat'][0].numpy().ravel(), z['item_feat'].ravel())[0, 1]
    assert r > 0.98 and ri > 0.98, (r, ri)
