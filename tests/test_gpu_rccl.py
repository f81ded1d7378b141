"""The person-sharded train step on a real RCCL communicator (backend 'nccl' = RCCL on ROCm), one rank on one GPU: a
1-rank all-reduce is the identity, so the sharded step must reproduce the unsharded one BITWISE -- through all three
ways bench.py / the CLI drive it:
  eager            forward_backward -> dist.all_reduce -> update
  two graphs       hipGraph(forward_backward) -> eager dist.all_reduce -> hipGraph(update)
  captured         one hipGraph with the collective recorded inside it
(The 8-GPU scaling run is the driver's; this pins that the RCCL path works on the hardware at all and that neither graph
variant changes a bit.  Runs in a subprocess: a process group cannot be re-initialised inside the pytest process.)"""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import copy, os, sys
    sys.path.insert(0, os.path.join(ROOT, 'variational-item-response-theory-public_amd')); sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from vibo_amd.torch_core.models import VIBO_2PL
    from vibo_amd.trainer import FusedTrainer
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    P, I, A, STEPS = 4099, 1000, 8, 6
    g = torch.Generator(device=dev).manual_seed(5)
    resp = (torch.rand(P, I, device=dev, generator=g) < 0.5).float()
    mask = torch.rand(P, I, device=dev, generator=g) >= 0.1
    torch.manual_seed(11)
    base = VIBO_2PL(A, I, ability_merge='product').to(dev)

    def run(mode):
        model = copy.deepcopy(base)
        sharded = mode != 'unsharded'
        if sharded:
            model.enable_person_sharding(lambda flat: dist.all_reduce(flat), seed=7, rank=0)
        tr = FusedTrainer(model, lr=5e-3, rng='native', seed=7)
        losses = []
        if mode in ('unsharded', 'eager'):
            for _ in range(STEPS):
                losses.append(float(tr.step(resp, mask)))
        else:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    losses.append(float(tr.step(resp, mask)))          # warm-up steps count as steps
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if mode == 'two_graphs':
                g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1):
                    raw = tr.forward_backward(resp, mask)
                with torch.cuda.graph(g2, pool=g1.pool()):
                    loss = tr.update()
                for _ in range(STEPS - 2):
                    g1.replay(); dist.all_reduce(raw.flat); g2.replay()
                    losses.append(float(loss))
            else:
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1):
                    loss = tr.step(resp, mask)
                for _ in range(STEPS - 2):
                    g1.replay()
                    losses.append(float(loss))
        torch.cuda.synchronize()
        return losses, {k: v.detach().clone() for k, v in model.state_dict().items()}

    ref_l, ref_s = run('unsharded')
    for mode in ('eager', 'two_graphs', 'captured'):
        l, s = run(mode)
        assert l == ref_l, (mode, l, ref_l)
        for k in ref_s:
            assert torch.equal(s[k], ref_s[k]), (mode, k)
        print('ok', mode)
    dist.destroy_process_group()
    print('RCCL_1RANK_OK')
''')


def test_one_rank_rccl_group_reproduces_the_unsharded_step_bitwise(tmp_path):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    script = tmp_path / 'rccl_1rank.py'
    script.write_text(f'ROOT = {ROOT!r}\n' + SCRIPT)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'RCCL_1RANK_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_bench_two_ranks_share_one_device_over_gloo(scaling):
    """bench.py's N > 1 path end to end -- torchrun, person shards per rank, the two-graph step around the collective, barriers,
    max-over-ranks timing, ONE JSON line from rank 0 -- on a single-GPU box: both ranks on cuda:0 with the gloo backend
    (VIBO_BENCH_ONE_DEVICE=1; RCCL refuses two ranks on one device).  The rates are meaningless; the contract is checked."""
    import json
    env = dict(os.environ, VIBO_BENCH_ONE_DEVICE='1', PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')]))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29571', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--scaling', scaling,
           '--persons', '100000']          # (otherwise the flags the driver passes: extras, `also` and Format P legs must not deadlock)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['warmup'] == 1 and d['scaling'] == scaling
    assert d['config']['global_batch'] == (200000 if scaling == 'weak' else 100000)
    assert d['value'] > 0 and d['ms_per_step'] > 0 and 'roofline' in d and 'cpu_baseline' not in d and 'extra' not in d
    assert 'also' in d and 'format_p' in d
    assert abs(d['value'] - d['config']['global_batch'] * 1000 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    # the per-phase breakdown a scaling run is diagnosed with: [noise + prologue + ELBO kernel + finalize | all-reduce | epilogue + Adam]
    ph = d['config']['phases_ms']
    assert ph is not None and all(ph[k] > 0 for k in ('forward_backward_graph', 'all_reduce', 'update_graph'))
    assert d['config']['persons_per_rank'] == (100000 if scaling == 'weak' else 50000)
