"""Host-side drop-in surface on CPU: CLI flags / out-dir naming / checkpoint layout
(vibo.py:24-142, 478-558), dataset split + artificial masking semantics
(datasets.py:46-78, 866-940), with the native entry points replaced by the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import cpu_backend
from vibo_amd import config, datasets, ops
from vibo_amd.torch_core import vibo as cli


@pytest.fixture()
def cpu_ops():
    restore = cpu_backend.install(ops)
    yield
    restore()


@pytest.fixture()
def tmp_dirs(tmp_path, monkeypatch):
    monkeypatch.setattr(config, 'DATA_DIR', str(tmp_path / 'data'))
    monkeypatch.setattr(config, 'OUT_DIR', str(tmp_path / 'out'))
    return tmp_path


def test_flags_and_defaults_match_reference():
    a = cli.build_parser().parse_args([])
    ref_defaults = dict(irt_model='1pl', dataset='1pl_simulation', ability_dim=1, ability_merge='product',
                        conditional_posterior=False, generative_model='irt', response_dist='bernoulli',
                        drop_missing=False, artificial_missing_perc=0., n_norm_flows=0, no_infer_dict=False,
                        no_marginal=False, no_test=False, no_predictive=False, num_person=1000, num_item=100,
                        num_posterior_samples=400, hidden_dim=64, max_num_person=None, max_num_item=None,
                        lr=5e-3, batch_size=16, epochs=100, max_iters=-1, num_workers=0, anneal_kl=False,
                        beta_kl=1.0, seed=42, gpu_device=0, cuda=False)
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, k


def test_flag_interactions_and_out_dir_name():
    a = cli.finalize_args(cli.build_parser().parse_args(
        ['--irt-model', '2pl', '--dataset', '2pl_simulation', '--n-norm-flows', '2', '--max-num-person', '7']))
    assert a.no_infer_dict and a.no_predictive and a.max_num_person is None
    a = cli.finalize_args(cli.build_parser().parse_args(
        ['--dataset', 'critlangacq', '--artificial-missing-perc', '0.2', '--no-predictive', '--max-num-person', '50']))
    assert a.no_predictive is False and a.num_person is None and a.max_num_person == 50
    assert cli.out_dir_name(a) == ('VIBO_1pl_critlangacq_bernoulli_irt_Noneperson_Noneitem_50maxperson_'
                                   'Nonemaxitem_0.2maskperc_1ability_product_unconditional_q_seed42')
    a = cli.finalize_args(cli.build_parser().parse_args(['--dataset', '3pl_simulation', '--irt-model', '3pl']))
    assert a.max_num_person is None      # the reference raises KeyError here (config.py:16-25); fixed


def test_simulation_split_and_getitem(tmp_dirs):
    tr = datasets.load_dataset('2pl_simulation', train=True, num_person=50, num_item=12, ability_dim=2)
    te = datasets.load_dataset('2pl_simulation', train=False, num_person=50, num_item=12, ability_dim=2)
    assert (tr.num_person, te.num_person, tr.num_item) == (40, 10, 12)
    assert os.path.exists(os.path.join(config.DATA_DIR, '2pl_simulation_50person_12item_2ability', 'simulation.pth'))
    idx, resp, item_id, mask = tr[3]
    assert resp.shape == (12, 1) and resp.dtype == torch.float32
    assert mask.dtype == torch.bool and item_id.dtype == torch.int64
    r, m = tr.matrix()
    assert r.shape == (40, 12) and m.all()


@pytest.mark.parametrize('name,sub', [('duolingo', 'duolingo'), ('wordbank', 'wordbankr'), ('pisa2015_science', 'pisa2015_science')])
def test_real_world_loaders_from_the_reference_cache_match_the_reference(tmp_dirs, name, sub):
    """Duolingo / WordBank / PISA (src/datasets.py:443-863) start from the score matrix the reference caches after parsing the raw
    corpus; datasets.CachedScoreMatrix starts from the same file.  Golden: the REAL reference loaders run on a synthetic cache
    (tools/gen_loader_golden.py -> tests/golden/score_matrix_loaders.npz): binarisation, RandomState(42) shuffle (not for
    WordBank), 80/20 split, max_num_person / max_num_item, all-missing rows dropped (not for WordBank), the per-sample tuple."""
    z = np.load(os.path.join(GOLDEN_DIR, 'score_matrix_loaders.npz'))
    d = os.path.join(config.DATA_DIR, sub)
    os.makedirs(d)
    np.save(os.path.join(d, 'score_matrix.npy'), z['in.duolingo'] if name == 'duolingo' else z['in.base'])
    if name == 'duolingo':
        np.save(os.path.join(d, 'token_id.npy'), z['in.token_id'])
    for tag, kw in (('train', dict(train=True)), ('test', dict(train=False)),
                    ('train_max', dict(train=True, max_num_person=30, max_num_item=11))):
        if name == 'wordbank' and 'max_num_item' in kw:
            kw = dict(train=True, max_num_person=30)      # (the reference's WordBank loader itself dies on max_num_item, datasets.py:686)
        ds = datasets.load_dataset(name, **kw)
        assert np.array_equal(ds.response, z[f'{name}.{tag}.response']) and np.array_equal(ds.mask, z[f'{name}.{tag}.mask']), tag
        assert (ds.num_person, ds.num_item) == z[f'{name}.{tag}.response'].shape and len(ds) == ds.num_person
        idx, r, iid, m = ds[2]
        assert idx == 2 and r.dtype == torch.float32 and iid.dtype == torch.int64 and m.dtype == torch.bool
        assert np.array_equal(r.numpy(), z[f'{name}.{tag}.item2.response'])
        assert np.array_equal(iid.numpy(), z[f'{name}.{tag}.item2.item_id']) and np.array_equal(m.numpy(), z[f'{name}.{tag}.item2.mask'])
        if f'{name}.{tag}.item_id' in z.files:
            assert np.array_equal(ds.item_id, z[f'{name}.{tag}.item_id'])
        rr, mm = ds.matrix()                  # what the resident split keeps in HBM
        assert rr.dtype == np.float32 and mm.dtype == bool and np.array_equal(mm, ds.response != -1)
    # WordBank with max_num_item: a NameError in the reference, simply the first items here
    if name == 'wordbank':
        assert datasets.load_dataset(name, train=True, max_num_item=5).num_item == 5


def test_score_matrix_dataset_trains_through_the_cli(cpu_ops, tmp_dirs):
    """--dataset score_matrix: any pre-built [P, I] matrix (1 right / 0 wrong / -1 missing) under DATA_DIR/score_matrix/ behind the
    reference's (index, response, item_id, mask) contract -- PISA's steps behind the cache -- runs end to end; without the file
    the loader names what is missing."""
    with pytest.raises(FileNotFoundError, match='score_matrix.npy'):
        datasets.load_dataset('score_matrix', train=True)
    rs = np.random.RandomState(3)
    M = rs.randint(0, 2, size=(120, 17)).astype(np.float32)
    M[rs.rand(120, 17) < 0.25] = -1
    M[5] = -1
    os.makedirs(os.path.join(config.DATA_DIR, 'score_matrix'))
    np.save(os.path.join(config.DATA_DIR, 'score_matrix', 'score_matrix.npy'), M)
    tr, te = datasets.load_dataset('score_matrix', train=True), datasets.load_dataset('score_matrix', train=False)
    assert tr.num_person + te.num_person == 119 and tr.num_item == 17          # (the all-missing row is gone)
    cli.main(['--irt-model', '2pl', '--dataset', 'score_matrix', '--epochs', '1', '--batch-size', '16', '--num-posterior-samples', '2',
              '--no-marginal', '--no-predictive', '--out-dir', config.OUT_DIR])
    (run_dir,) = os.listdir(config.OUT_DIR)
    assert run_dir.startswith('VIBO_2pl_score_matrix_bernoulli_irt_Noneperson_Noneitem')
    ck = torch.load(os.path.join(config.OUT_DIR, run_dir, 'checkpoint.pth.tar'), weights_only=False)
    assert ck['infer_dict']['ability_mu'].shape == (tr.num_person, 1)


def test_artificial_mask_matches_reference_semantics(tmp_dirs):
    """Golden from the reference's artificially_mask_dataset on a 200 x 95 matrix (perc 0.2)."""
    z = np.load(os.path.join(GOLDEN_DIR, 'artificial_mask.npz'))

    class D:
        pass
    d = D()
    d.response = z['response'].copy()
    d.mask = z['mask'].copy()
    out = datasets.artificially_mask_dataset(d, 0.2)
    assert np.array_equal(out.missing_indices, z['missing_indices'])
    assert np.array_equal(np.asarray(out.missing_labels).reshape(-1), z['missing_labels'].reshape(-1))
    assert np.array_equal(out.mask, z['masked_mask']) and np.array_equal(out.response, z['masked_response'])
    assert d.mask.all()                                   # the input dataset is not modified


@pytest.mark.parametrize('extra', [[], ['--artificial-missing-perc', '0.2', '--conditional-posterior'],
                                   ['--n-norm-flows', '2', '--irt-model', '3pl', '--dataset', '3pl_simulation']])
def test_cli_end_to_end_checkpoint_layout(cpu_ops, tmp_dirs, extra):
    argv = ['--irt-model', '2pl', '--dataset', '2pl_simulation', '--num-person', '300', '--num-item', '12',
            '--epochs', '4', '--batch-size', '16', '--num-posterior-samples', '3',
            '--out-dir', str(tmp_dirs / 'out')] + extra
    cli.main(argv)
    (run_dir,) = os.listdir(tmp_dirs / 'out')
    files = set(os.listdir(tmp_dirs / 'out' / run_dir))
    assert {'checkpoint.pth.tar', 'model_best.pth.tar', 'train_losses.npy', 'test_losses.npy',
            'train_times.npy'} <= files
    ck = torch.load(tmp_dirs / 'out' / run_dir / 'checkpoint.pth.tar', weights_only=False)
    assert {'model_state_dict', 'epoch', 'args', 'train_logp', 'test_logp'} <= set(ck)
    flows = '--n-norm-flows' in extra
    if not flows:
        assert set(ck['infer_dict']) == {'ability_mu', 'ability_logvar', 'item_feat_mu', 'item_feat_logvar'}
        assert ck['infer_dict']['ability_mu'].shape == (240, 1)
        assert ck['posterior_predict_samples']['response'].shape[1:] == (240, 12, 1)
    if '--artificial-missing-perc' in extra:
        assert 0.0 <= ck['missing_imputation_accuracy'] <= 1.0
        assert 0.0 <= ck['missing_imputation_accuracy_mean'] <= 1.0
    losses = np.load(tmp_dirs / 'out' / run_dir / 'train_losses.npy')
    times = np.load(tmp_dirs / 'out' / run_dir / 'train_times.npy')
    assert losses.shape == (4,) and np.isfinite(losses).all() and (times <= 0).all()
    assert losses[-1] < losses[0]                         # it learns


@pytest.mark.parametrize('extra', [[], ['--artificial-missing-perc', '0.2', '--irt-model', '3pl', '--dataset', '3pl_simulation',
                                        '--ability-dim', '2']])
def test_vi_cli_end_to_end_checkpoint_layout(cpu_ops, tmp_dirs, extra):
    """Drop-in for src/torch_core/vi.py (flags vi.py:20-78, out-dir name :87-96, checkpoint keys :318-368)."""
    from vibo_amd.torch_core import vi
    argv = ['--irt-model', '2pl', '--dataset', '2pl_simulation', '--num-person', '300', '--num-item', '12',
            '--epochs', '5', '--batch-size', '16', '--num-posterior-samples', '3', '--lr', '0.02',
            '--out-dir', str(tmp_dirs / 'out')] + extra
    out_dir = vi.main(argv)
    (run_dir,) = os.listdir(tmp_dirs / 'out')
    irt = '3pl' if extra else '2pl'
    assert run_dir == f'vi_{irt}_{irt}_simulation_300person_12item_{0.2 if extra else 0.0}maskperc_{2 if extra else 1}ability'
    assert out_dir.endswith(run_dir)
    files = set(os.listdir(tmp_dirs / 'out' / run_dir))
    assert {'checkpoint.pth.tar', 'model_best.pth.tar', 'train_losses.npy', 'train_times.npy'} <= files
    ck = torch.load(tmp_dirs / 'out' / run_dir / 'checkpoint.pth.tar', weights_only=False)
    assert {'model_state_dict', 'epoch', 'args', 'train_logp', 'infer_dict', 'posterior_predict_samples'} <= set(ck)
    assert set(ck['model_state_dict']) == {'ability_mu_lookup.weight', 'ability_logvar_lookup.weight', 'item_mu_lookup.weight',
                                           'item_logvar_lookup.weight'}
    assert ck['infer_dict']['ability_mu'].shape == (240, 2 if extra else 1)           # the train split: first 80 % of the persons
    losses = np.load(tmp_dirs / 'out' / run_dir / 'train_losses.npy')
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    if extra:
        assert 0.0 <= ck['missing_imputation_accuracy'] <= 1.0


def test_posthoc_scripts_enrich_a_checkpoint(cpu_ops, tmp_dirs):
    """infer.py / marginal.py / predictives.py drop-ins: rebuild everything from the checkpoint's own args and write the
    result back into the file."""
    from vibo_amd.torch_core import infer, marginal, predictives
    cli.main(['--irt-model', '2pl', '--dataset', '2pl_simulation', '--num-person', '300', '--num-item', '12', '--epochs', '2',
              '--batch-size', '16', '--num-posterior-samples', '3', '--artificial-missing-perc', '0.2', '--no-infer-dict',
              '--no-marginal', '--no-test', '--out-dir', str(tmp_dirs / 'out')])
    (run_dir,) = os.listdir(tmp_dirs / 'out')
    path = str(tmp_dirs / 'out' / run_dir / 'checkpoint.pth.tar')
    before = torch.load(path, weights_only=False)
    assert 'infer_dict' not in before and 'train_logp' not in before
    ck = infer.main([path])
    assert ck['infer_dict']['ability_mu'].shape == (240, 1)
    ck = marginal.main([path])
    assert np.isfinite(ck['train_logp']) and np.isfinite(ck['test_logp'])
    ck = predictives.main([path, '--num-posterior-samples', '5'])
    assert ck['posterior_predict_samples']['response'].shape[1:3] == (240, 12) and 0.0 <= ck['missing_imputation_accuracy'] <= 1.0
    after = torch.load(path, weights_only=False)
    assert {'infer_dict', 'train_logp', 'test_logp', 'posterior_predict_samples', 'missing_imputation_accuracy'} <= set(after)
    assert after['args'].num_posterior_samples == 5
