"""Person-sharded data parallelism with 2 processes over gloo on CPU:
each rank holds half of the persons, ONE all-reduce (sum) of the kernel's flat
[scalars | grads] buffer per step; loss and every parameter gradient must equal
the single-process result on the whole batch (here: the reference golden).
The native entry point is replaced by the CPU oracle (tests only); on the GPU box
the same code path runs with backend 'nccl' (= RCCL over xGMI) in bench.py / the CLI."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN_DIR, Golden, rel_err

CASES = ['2pl_a8_uncond_miss_prior', '3pl_a1_cond_miss_drop', '2pl_a2_uncond_flows2_miss', '2pl_a2_uncond_mean_miss',
         # MLP decoders (models.py:769-919): autograd on the rank's persons + ONE flat gradient all-reduce (allreduce_grads)
         '2pl_a2_deep_miss', '1pl_a3_residual_mean_miss_drop', '3pl_a2_link_flows2', '2pl_a2_cond_residual_miss']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import cpu_backend
        from test_host_logic import build_model
        from vibo_amd import ops
        cpu_backend.install(ops)
        g = Golden(os.path.join(GOLDEN_DIR, f'case_{case}.npz'))
        m = g.meta
        model = build_model(g)
        calls = []

        def reducer(flat):
            calls.append(flat.numel())
            dist.all_reduce(flat)
        model.enable_person_sharding(reducer, seed=0, rank=rank)
        B = m['num_person']
        lo, hi = rank * B // world, (rank + 1) * B // world
        resp, mask = g.response[lo:hi].unsqueeze(2), g.mask[lo:hi].long().unsqueeze(2)
        outs = model(resp, mask, eps_item=g.eps_item, eps_ability=g.eps_ability[lo:hi])
        if m['n_norm_flows'] > 0:
            (r, k, rmu, ak, a0, amu, alv, aladj, ik, i0, imu, ilv, iladj) = outs
            loss = model.elbo(r, k, rmu, a0, amu, alv, i0, imu, ilv, annealing_factor=m['annealing_factor'],
                              use_kl_divergence=False, ability_k=ak, item_feat_k=ik,
                              ability_logabsdetjac=aladj, item_logabsdetjac=iladj)
        else:
            loss = model.elbo(*outs, annealing_factor=m['annealing_factor'],
                              use_kl_divergence=m['use_kl_divergence'])
        loss.backward()
        if m.get('generative_model', 'irt') != 'irt':
            assert model.needs_grad_allreduce and len(calls) == 0
            loss = model.allreduce_grads(loss)
            assert len(calls) == 1, 'exactly one collective per step'
        elif m.get('ability_merge', 'product') == 'mean':
            # the per-person posterior gradients stay local; scalars and item gradients are reduced in the forward,
            # the 8 small encoder tensors (mlp1 features, mlp2 weights / biases) in the backward
            assert 2 <= len(calls) <= 8 and max(calls) <= 64 * 64 + 2 * g.grad['item_encoder.mu_lookup.weight'].numel()
        else:
            assert len(calls) == 1, 'exactly one collective per step'
        torch.save({'loss': loss.detach(), 'grads': {n: p.grad for n, p in model.named_parameters()},
                    'ncalls': len(calls)}, f'{out_path}.{rank}')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case', CASES)
def test_two_rank_shard_sum_equals_single_process(case, tmp_path):
    world = 2
    out = str(tmp_path / 'res')
    mp.spawn(_worker, args=(world, _free_port(), case, out), nprocs=world, join=True)
    g = Golden(os.path.join(GOLDEN_DIR, f'case_{case}.npz'))
    res = [torch.load(f'{out}.{r}', weights_only=False) for r in range(world)]
    # every rank sees the same global loss and the same gradients (replicas stay in lock-step)
    assert float(res[0]['loss']) == float(res[1]['loss'])
    assert rel_err(res[0]['loss'], g.out['loss']) < 1e-4
    for name, gref in g.grad.items():
        a, b = res[0]['grads'][name], res[1]['grads'][name]
        assert torch.equal(a, b), name
        if float(gref.abs().max()) > 0:
            assert rel_err(a, gref) < 1.5e-3, name        # fp32 CPU stand-in; the reference itself is this noisy on 3PL


def _cli_worker(rank, world, port, data_dir, out_dir, num_person):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    from oracle import cpu_backend
    from vibo_amd import config, ops
    from vibo_amd.torch_core import vibo as cli
    cpu_backend.install(ops)
    config.DATA_DIR, config.OUT_DIR = data_dir, out_dir
    cli.main(['--irt-model', '2pl', '--dataset', '2pl_simulation', '--num-person', str(num_person), '--num-item', '12',
              '--epochs', '2', '--batch-size', '16', '--num-posterior-samples', '2', '--no-marginal', '--no-predictive',
              '--out-dir', out_dir])


def test_cli_two_ranks_with_uneven_shards_takes_equal_steps(tmp_path):
    """1002 persons -> 801 train rows -> shards of 400 and 401 rows at 8 rows per rank and step: 50 vs 51 minibatches if
    every rank cut its own shard by batch size, i.e. one rank would issue an all-reduce its peer never joins (a hang).
    Both ranks must take the same number of steps and finish."""
    world, num_person = 2, 1002
    data_dir, out_dir = str(tmp_path / 'data'), str(tmp_path / 'out')
    ctx = mp.spawn(_cli_worker, args=(world, _free_port(), data_dir, out_dir, num_person), nprocs=world, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=5):
        assert time.time() - t0 < 240, 'person-sharded CLI run did not finish: ranks disagree on the number of collectives'
    (run_dir,) = os.listdir(out_dir)
    ck = torch.load(os.path.join(out_dir, run_dir, 'checkpoint.pth.tar'), weights_only=False)
    assert ck['infer_dict']['ability_mu'].shape[0] == 801
    # the checkpoint's infer_dict was computed by the two ranks on their own shards (400 + 401 rows) and all-gathered in rank
    # order: it has to be what one process computes on the whole split from the same checkpoint
    from oracle import cpu_backend
    from vibo_amd import config, ops
    from vibo_amd.torch_core import posthoc, vibo as cli
    restore = cpu_backend.install(ops)
    saved = config.DATA_DIR, config.OUT_DIR
    try:
        config.DATA_DIR, config.OUT_DIR = data_dir, out_dir
        args = ck['args']
        ds = posthoc._dataset(args, True)
        model = posthoc._model(args, ds.num_item, ck['model_state_dict'], torch.device('cpu'))
        whole = cli.infer_dict(model, posthoc._split(args, ds, torch.device('cpu')), args.batch_size)
    finally:
        config.DATA_DIR, config.OUT_DIR = saved
        restore()
    for k in ('ability_mu', 'ability_logvar', 'item_feat_mu', 'item_feat_logvar'):
        assert torch.allclose(ck['infer_dict'][k].cpu(), whole[k].cpu(), rtol=1e-6, atol=1e-6), k
