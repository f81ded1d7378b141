#!/usr/bin/env python3
"""bench.py -- person x item ELBO terms/sec of the fused VIBO train step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json: "person x item ELBO terms/sec on 1M x 1k 2PL", configs[2]'s shape per GPU):
2PL, 1 000 000 persons x 1 000 items per GPU, ability_dim 8, synthetic Bernoulli responses with 10 % missing
cells, device-resident (inputs are in HBM before the timed region).  `also` repeats the measurement at
ability_dim 1 (configs[1]'s width, the reference default).  One step = one ELBO train step over the GPU's whole
person shard, replayed from a hipGraph: reparameterisation noise (vibo_fill_normal), vibo_train_prologue (item
sample, item KL, encoder table), the fused HIP forward+backward over the [B,I] response matrix + finalize, ONE
all-reduce of the flat [scalars|grads] buffer when N > 1 (persons are sharded, weak scaling; two graphs around an
eager RCCL all-reduce), vibo_train_epilogue (loss, encoder-MLP / item backward, Adam).
--torch-optimizer runs the O(I) part as PyTorch autograd + torch.optim.Adam instead.

Adds to the contract line:
  roofline      the fused kernel's achieved HBM GB/s = algorithmic bytes (5 + 12A/I per term) x terms
                per launch / its average duration, measured with HIP events on the launch stream (an eager pass of
                the same step right after the timed region when the step is graph-replayed); peak 8000 GB/s
                (MI355X_MICROARCH.md), also as a fraction of the 6290 GB/s measured copy ceiling; traffic = recorded
                2*FETCH_SIZE + WRITE_SIZE of the same command (profiles/r01_bench_profile.txt).
  cpu_baseline  the CPU oracle port of the reference op sequence (per-term MLP -> PoE -> link -> masked log-lik ->
                autograd -> Adam, oracle/vibo_oracle.py) timed on this host's cores on a bounded sample (rank 0,
                N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, 'variational-item-response-theory-public_amd')
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--persons', type=int, default=1_000_000, help='persons per GPU (weak scaling)')
    ap.add_argument('--items', type=int, default=1000)
    ap.add_argument('--ability-dim', type=int, default=8)
    ap.add_argument('--also-ability-dim', type=int, default=1, help='second workload reported under "also" (0 = none)')
    ap.add_argument('--irt-model', type=str, default='2pl', choices=['1pl', '2pl', '3pl'])
    ap.add_argument('--missing', type=float, default=0.1)
    ap.add_argument('--lr', type=float, default=5e-3)
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=2048)
    ap.add_argument('--cpu-steps', type=int, default=8)
    ap.add_argument('--eval-only', action='store_true', help='forward ELBO only (no backward/optimizer)')
    ap.add_argument('--ability-merge', choices=['product', 'mean'], default='product', help="'mean': the reference's other encoder (models.py:631-650) through VIBO_POSTERIOR_GIVEN + torch autograd / Adam (implies --torch-optimizer --no-graph)")
    ap.add_argument('--no-format-p', action='store_true', help='skip the extra Format P (1-byte cell codes) measurement of the same step')
    ap.add_argument('--torch-optimizer', action='store_true', help='PyTorch autograd + torch.optim.Adam for the O(I) part instead of the fused prologue/epilogue kernels')
    ap.add_argument('--rng', choices=['native', 'torch'], default='native', help='reparameterisation noise: vibo_fill_normal (Philox, in the C ABI) or torch.randn')
    ap.add_argument('--no-graph', action='store_true', help='launch every step eagerly instead of replaying a hipGraph')
    ap.add_argument('--graph-collective', action='store_true', help='multi-GPU: capture the all-reduce inside the step graph instead of two graphs around an eager all-reduce')
    ap.add_argument('--force-dist', action='store_true', help='create the process group even for one rank (tests the RCCL path)')
    return ap.parse_args()


def synth_responses(irt, P, I, A, missing, device, seed):
    """theta ~ N(0,1), item ~ N(0,1), r ~ Bernoulli(link) (src/pyro_core/models.py:68-110
    semantics), generated on the device in chunks; -1 / mask 0 where missing."""
    g = torch.Generator(device=device).manual_seed(seed)
    D = {1: 1, 2: A + 1, 3: A + 2}[irt]
    item = torch.randn(I, D, device=device, generator=g)
    resp = torch.empty(P, I, dtype=torch.float32, device=device)
    mask = torch.empty(P, I, dtype=torch.bool, device=device)
    chunk = 65536
    for s in range(0, P, chunk):
        n = min(chunk, P - s)
        theta = torch.randn(n, A, device=device, generator=g)
        if irt == 1:
            logit = theta.sum(1, keepdim=True) + item[:, 0]
        else:
            logit = -(theta @ item[:, :A].t()) + item[:, A]
        p = torch.sigmoid(logit)
        if irt == 3:
            gs = torch.sigmoid(item[:, A + 1])
            p = gs + (1 - gs) * p
        r = torch.bernoulli(p, generator=g)
        m = torch.rand(n, I, device=device, generator=g) >= missing
        resp[s:s + n] = torch.where(m, r, torch.full_like(r, -1.0))
        mask[s:s + n] = m
    return resp, mask


def cpu_baseline(args, irt):
    """Reference op sequence on the host cores (oracle port), train step, bounded sample."""
    from oracle import vibo_oracle as O
    A, I, B = args.ability_dim, args.items, args.cpu_batch
    g = torch.Generator().manual_seed(args.seed)
    resp, mask = O.simulate_responses(irt, B, I, A, generator=g, missing_frac=0.0)
    params = {k: v.requires_grad_(True) for k, v in O.init_params(irt, A, I, generator=g).items()}
    opt = torch.optim.Adam(list(params.values()), lr=args.lr)

    def step():
        opt.zero_grad()
        out = O.elbo_forward(params, resp, mask, torch.randn(I, O.item_feat_dim(irt, A)), torch.randn(B, A),
                             irt_model=irt, ability_dim=A)
        out['loss'].backward()
        opt.step()

    # the per-term MLP is many mid-sized ops: on a many-core host the default thread count (all cores) is slower than a
    # moderate one, so probe a few and time the best -- the baseline should be the CPU path at its best
    all_threads = torch.get_num_threads()
    step()
    best, best_dt = all_threads, None
    for nt in sorted({min(all_threads, n) for n in (8, 16, 32, 64, all_threads)}):
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter()
        step()
        d1 = time.perf_counter() - t0
        if best_dt is None or d1 < best_dt:
            best, best_dt = nt, d1
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        step()
    dt = time.perf_counter() - t0
    torch.set_num_threads(all_threads)
    return {
        'value': B * I * args.cpu_steps / dt, 'unit': 'terms/s', 'cores': best,
        'kind': 'port', 'host_threads_available': all_threads,
        'sample': f'{args.cpu_steps} train steps of {B} persons x {I} items (ability_dim {A}, no missing), '
                  f'oracle/vibo_oracle.py = reference op sequence incl. per-term encoder MLP, autograd, Adam; '
                  f'thread count = best of a short probe over 8..{all_threads}',
    }


def main():
    args = parse()
    if args.ability_merge == 'mean':
        args.torch_optimizer, args.no_graph, args.also_ability_dim = True, True, 0
    irt = int(args.irt_model[0])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run for --gpus > 1')
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from vibo_amd import ops
    from vibo_amd.torch_core.models import VIBO_1PL, VIBO_2PL, VIBO_3PL

    def measure(A, codes=False):
        """-> dict(dt, kern_ms, final_loss, graph) for ability_dim A on this rank's shard."""
        P, I = args.persons, args.items
        resp, mask = synth_responses(irt, P, I, A, args.missing, dev, args.seed + 1000 * rank)
        if codes:           # Format P: the same matrix as one byte per cell (VIBO_MASK_CODES)
            resp, mask = ops.pack_cell_codes(resp, mask), None
            torch.cuda.empty_cache()
        torch.manual_seed(args.seed)
        model = {1: VIBO_1PL, 2: VIBO_2PL, 3: VIBO_3PL}[irt](A, I, ability_merge=args.ability_merge).to(dev)
        if dist is not None:
            model.enable_person_sharding(lambda flat: dist.all_reduce(flat), seed=args.seed, rank=rank)
        trainer, opt = None, None
        if args.torch_optimizer or args.eval_only:
            opt = torch.optim.Adam(model.parameters(), lr=args.lr, capturable=not args.no_graph, fused=True)
        else:
            from vibo_amd.trainer import FusedTrainer
            trainer = FusedTrainer(model, lr=args.lr, rng=args.rng, seed=args.seed)

        # HIP events around the native call, on the stream it is launched on
        events = []
        native = ops._BACKEND['elbo']
        recording = {'on': False}

        def timed_native(*a, **k):
            if not recording['on']:
                return native(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = native(*a, **k)
            e1.record()
            events.append((e0, e1))
            return out

        ops._BACKEND['elbo'] = timed_native

        graph = None

        def step():
            if args.eval_only:
                with torch.no_grad():
                    return model.elbo_step(resp, mask)
            if trainer is not None:
                return trainer.step(resp, mask)
            opt.zero_grad(set_to_none=False)
            loss = model.elbo_step(resp, mask)
            loss.backward()
            opt.step()
            return loss.detach()

        # The whole step (PyTorch O(I) part, fused HIP kernel, all-reduce, autograd, Adam) is captured once into a
        # hipGraph and replayed: ~80 tiny launches per step cost more than a quarter of the 1-2 ms kernel otherwise.
        eager_step = step
        if not args.no_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        step()                   # allocator / hipFuncSetAttribute / Adam state / RCCL warm-up outside capture
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                for gen in (model._item_gen, model._ability_gen):      # dedicated generators of the person-sharded mode
                    if gen is not None:
                        g.register_generator_state(gen)
                if opt is not None:
                    opt.zero_grad(set_to_none=False)
                if dist is not None and trainer is not None and not args.graph_collective:
                    # person-sharded: two graphs around an EAGER all-reduce (a collective inside a captured graph
                    # is one more thing that can go wrong on a node this script has never run on)
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        static_raw = trainer.forward_backward(resp, mask)
                    with torch.cuda.graph(g2, pool=g.pool()):
                        static_loss = trainer.update()
                    graph = g

                    def step():
                        graph.replay()
                        dist.all_reduce(static_raw.flat)
                        g2.replay()
                        return static_loss
                else:
                    with torch.cuda.graph(g):
                        static_loss = step()
                    graph = g

                    def step():
                        graph.replay()
                        return static_loss
            except Exception as exc:             # never lose the measurement to a capture problem
                print(f'[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); running eagerly', file=sys.stderr)
                graph = None
                step = eager_step
                torch.cuda.synchronize()

        for _ in range(args.warmup):
            loss = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        recording['on'] = graph is None
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        recording['on'] = False
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        final_loss = float(loss.detach())

        if graph is not None:
            # events cannot be recorded inside a replayed graph: time the native call on the same stream / inputs
            # in an eager pass of the same step right after the timed region
            recording['on'] = True
            for _ in range(min(args.steps, 10)):
                eager_step()
            torch.cuda.synchronize()
            recording['on'] = False
        kern_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, len(events))
        bytes_per_term = 5.0 + 12.0 * A / I
        achieved = bytes_per_term * P * I / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        ops._BACKEND['elbo'] = native
        del resp, mask, model, opt, trainer
        torch.cuda.empty_cache()
        return dict(dt=dt, kern_ms=kern_ms, final_loss=final_loss, graph=graph is not None)

    P, I, A = args.persons, args.items, args.ability_dim
    m = measure(A)
    dt, kern_ms, final_loss = m['dt'], m['kern_ms'], m['final_loss']
    bytes_per_term = 5.0 + 12.0 * A / I
    achieved = bytes_per_term * P * I / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    also = None
    if args.also_ability_dim and args.also_ability_dim != A:
        A2 = args.also_ability_dim
        m2 = measure(A2)
        b2 = 5.0 + 12.0 * A2 / I
        also = {'workload': f'same, ability_dim={A2} (BASELINE configs[1] shape at 1M persons)',
                'value': float(P) * I * args.steps * world / m2['dt'], 'ms_per_step': m2['dt'] / args.steps * 1e3,
                'kernel_ms': m2['kern_ms'], 'roofline_achieved_GBps': b2 * P * I / (m2['kern_ms'] * 1e-3) / 1e9,
                'roofline_frac': b2 * P * I / (m2['kern_ms'] * 1e-3) / 1e9 / 8000.0,
                'roofline_frac_of_measured_copy_peak': b2 * P * I / (m2['kern_ms'] * 1e-3) / 1e9 / 6290.0}

    format_p = None
    if not args.no_format_p:
        m3 = measure(A, codes=True)
        b3 = 1.0 + 12.0 * A / I
        format_p = {'workload': 'same matrix and step as the headline, rows stored as 1-byte cell codes (Format P, VIBO_MASK_CODES): '
                                'reported separately, its own bytes figure, never mixed with the headline roofline',
                    'value': float(P) * I * args.steps * world / m3['dt'], 'unit': 'terms/s', 'ms_per_step': m3['dt'] / args.steps * 1e3,
                    'kernel_ms': m3['kern_ms'], 'bytes_per_term': b3,
                    'roofline_achieved_GBps': b3 * P * I / (m3['kern_ms'] * 1e-3) / 1e9,
                    'roofline_frac': b3 * P * I / (m3['kern_ms'] * 1e-3) / 1e9 / 8000.0,
                    'bound': 'valu (latency / issue), not hbm', 'final_loss_per_term': m3['final_loss'] / (P * I * world)}

    # HBM bytes per launch from rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction of
    # MI355X_MICROARCH.md) cannot be collected from inside this process; for the default workload the value
    # recorded in profiles/r01_bench_profile.txt is reported, otherwise null.
    traffic, traffic_note = None, 'not measured for this workload'
    recorded = {8: (2 * 2461234 + 111894) * 1024.0, 1: (2 * 2453304 + 15751) * 1024.0}     # KiB counters -> bytes
    recorded_p = {8: (2 * 508101 + 111894) * 1024.0}                                        # Format P launch (cell codes)
    if (P, I, irt, abs(args.missing - 0.1) < 1e-9) == (1_000_000, 1000, 2, True) and A in recorded:
        traffic = recorded[A]
        traffic_note = 'recorded measurement: profiles/r01_bench_profile.txt (2*FETCH_SIZE + WRITE_SIZE of vibo::split_kernel, KiB)'
    if also is not None and (P, I, irt, abs(args.missing - 0.1) < 1e-9) == (1_000_000, 1000, 2, True) and args.also_ability_dim in recorded:
        also['traffic'] = recorded[args.also_ability_dim]
    if format_p is not None and (P, I, irt, abs(args.missing - 0.1) < 1e-9) == (1_000_000, 1000, 2, True) and A in recorded_p:
        format_p['traffic'] = recorded_p[A]
    if rank == 0:
        terms = float(P) * I * args.steps * world
        line = {
            'metric': 'person x item ELBO terms/sec (train step: fwd + bwd + all-reduce + Adam)'
                      if not args.eval_only else 'person x item ELBO terms/sec (forward ELBO only)',
            'value': terms / dt, 'unit': 'terms/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.irt_model.upper()} simulation, {P} persons x {I} items per GPU, '
                                   f'ability_dim={A}, {args.missing:.0%} missing, {"product-of-experts" if args.ability_merge == "product" else "mean-merge"} encoder, '
                                   f'unconditional posterior, full-shard minibatch',
                       'global_batch': P * world, 'parallelism': f'person-sharded dp{world}', 'launch': ('hipGraph replay' if (dist is None or args.graph_collective) else 'two hipGraphs around an eager RCCL all-reduce') if m['graph'] else 'eager',
                       'optimizer': 'torch.optim.Adam (fused)' if (args.torch_optimizer or args.eval_only) else 'fused prologue/epilogue HIP kernels (Adam)',
                       'noise': 'torch.randn' if (args.torch_optimizer or args.eval_only or args.rng == 'torch') else 'Philox4x32-10 drawn in the prologue kernel (vibo_train_prologue_noise = the vibo_fill_normal streams)',
                       'final_loss_per_term': final_loss / (P * I * world)},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': achieved / 8000.0, 'frac_of_measured_copy_peak': achieved / 6290.0,
                         'traffic': traffic, 'traffic_note': traffic_note,
                         'kernel': 'vibo::split_kernel (+ item_prep, finalize helpers inside the timed events)',
                         'kernel_ms': kern_ms, 'bytes_per_term': bytes_per_term},
        }
        if also is not None:
            line['also'] = also
        if format_p is not None:
            line['format_p'] = format_p
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args, irt)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
