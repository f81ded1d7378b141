#!/usr/bin/env python3
"""Is the CPU stand-in that bench.py times (oracle/vibo_oracle.py, the reference op sequence) as fast as the REAL
reference on the same cores?  (within +-10 %.)

Build container only: imports /root/reference (never shipped) next to the port and times one train step of both --
forward, elbo, backward, Adam -- on identical shapes, thread counts and data, B = 16 and B = 1024, 3 warm-ups + >= 20
timed steps.  Prints both and their ratio; bench.py carries the ratio measured here as `validated_ratio`.

    python tools/validate_cpu_port.py [--items 1000] [--ability-dim 8] [--steps 20] [--threads 8]
"""
import argparse
import json
import os
import sys
import time
import types

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import torch

REF = '/root/reference'


def reference_step_fn(irt, A, I, B, lr, seed):
    sys.modules.setdefault('nltk', types.SimpleNamespace(word_tokenize=None))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    torch.distributions.Distribution.set_default_validate_args(False)
    from src.torch_core import models as M
    torch.manual_seed(seed)
    cls = {1: M.VIBO_1PL, 2: M.VIBO_2PL, 3: M.VIBO_3PL}[irt]
    model = cls(A, I, hidden_dim=64, ability_merge='product', conditional_posterior=False, generative_model='irt',
                response_dist='bernoulli', replace_missing_with_prior=True, n_norm_flows=0)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    g = torch.Generator().manual_seed(seed)
    resp = (torch.rand(B, I, 1, generator=g) < 0.5).float()
    mask = torch.ones(B, I, 1, dtype=torch.long)           # the train loop passes mask.long() (vibo.py:240)

    def step():
        opt.zero_grad()
        out = model(resp, mask)
        loss = model.elbo(*out, annealing_factor=1.0, use_kl_divergence=True)
        loss.backward()
        opt.step()
        return float(loss)
    return step


def port_step_fn(irt, A, I, B, lr, seed):
    from oracle import vibo_oracle as O
    g = torch.Generator().manual_seed(seed)
    resp = (torch.rand(B, I, generator=g) < 0.5).float()
    mask = torch.ones(B, I, dtype=torch.long)              # as the reference's loop hands it over (vibo.py:240)
    params = {k: v.requires_grad_(True) for k, v in O.init_params(irt, A, I, generator=g).items()}
    opt = torch.optim.Adam(list(params.values()), lr=lr)
    D = O.item_feat_dim(irt, A)

    def step():
        opt.zero_grad()
        out = O.elbo_forward(params, resp, mask, torch.randn(I, D), torch.randn(B, A), irt_model=irt, ability_dim=A)
        out['loss'].backward()
        opt.step()
        return float(out['loss'])
    return step


def time_interleaved(step_a, step_b, warm, steps, rounds=10):
    """The container's host is shared: alternate short blocks of the two step functions and keep each one's FASTEST block
    (interference only ever slows a block down)."""
    for _ in range(warm):
        step_a(); step_b()
    per = max(1, steps // rounds)
    ta, tb = [], []
    for _ in range(rounds):
        for fn, acc in ((step_a, ta), (step_b, tb)):
            t0 = time.perf_counter()
            for _ in range(per):
                fn()
            acc.append((time.perf_counter() - t0) / per)
    return min(ta), min(tb)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--items', type=int, default=1000)
    ap.add_argument('--ability-dim', type=int, default=8)
    ap.add_argument('--irt', type=int, default=2)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--threads', type=int, default=0, help='0 = torch default (all cores)')
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    res = {'threads': torch.get_num_threads(), 'items': a.items, 'ability_dim': a.ability_dim}
    for B in (16, 1024):
        steps = a.steps * 4 if B == 16 else a.steps
        t_ref, t_port = time_interleaved(reference_step_fn(a.irt, a.ability_dim, a.items, B, 5e-3, 42),
                                         port_step_fn(a.irt, a.ability_dim, a.items, B, 5e-3, 42), 3, steps)
        res[f'b{B}'] = {'reference_ms': t_ref * 1e3, 'port_ms': t_port * 1e3,
                        'reference_Mterms_s': B * a.items / t_ref / 1e6, 'port_Mterms_s': B * a.items / t_port / 1e6,
                        'ratio_port_over_reference_throughput': t_ref / t_port}
        print(f'B={B:5d}: reference {t_ref * 1e3:8.2f} ms/step ({B * a.items / t_ref / 1e6:5.2f} M terms/s)   '
              f'port {t_port * 1e3:8.2f} ms/step ({B * a.items / t_port / 1e6:5.2f} M terms/s)   '
              f'port/reference throughput = {t_ref / t_port:.3f}')
    print(json.dumps(res))


if __name__ == '__main__':
    main()
