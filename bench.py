#!/usr/bin/env python3
"""bench.py -- person x item ELBO terms/sec of the fused VIBO train step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json: "person x item ELBO terms/sec on 1M x 1k 2PL", configs[2] literally): 2PL, 1 000 000 persons x 1 000
items, ability_dim 8, synthetic Bernoulli responses with 10 % missing cells, device-resident (inputs are in HBM before the
timed region), the persons sharded over the N ranks (`--scaling strong`, the default: at N = 1 the one GPU holds the whole
matrix; `--scaling weak` gives every rank 1M persons, and a run with N > 1 reports that too, under `also_weak`).  `also`
repeats the measurement at ability_dim 1 (configs[1]'s width, the reference default).  One step = one ELBO train step over the
rank's whole person shard, replayed from a hipGraph -- TWO launches (the folded step, vibo_amd/trainer.py):
vibo_elbo_fwd_bwd_step (the row-split ELBO kernel; everything it reads -- item sample, encoder table, noise -- was left in
memory by the previous step's epilogue) and vibo_train_epilogue_fused (finalize, loss, encoder-MLP / item backward, Adam, and
the NEXT step's head: Philox noise, item sample, item KL, encoder table); with
N > 1 the finalize stays with the first launch and ONE all-reduce of the flat [scalars | grads] buffer sits between the two
(captured inside the step's graph by default, `--two-graphs` for an eager collective between two graphs).
--torch-optimizer runs the O(I) part as PyTorch autograd + torch.optim.Adam instead.

Adds to the contract line:
  roofline      the fused kernel's achieved HBM GB/s = algorithmic bytes (5 + 12A/I per term) x terms per launch
                / its mean duration INSIDE the replayed step: events cannot be recorded inside a replayed graph, so right after
                the timed region the same step is replayed as two hipGraphs (fused ELBO call | epilogue) with a HIP event on the
                launch stream before and after the first -- the kernel in the regime of the measured steps (an upper bound: the
                event packets cost a few us), the number the committed rocprofv3 kernel-trace average of this command
                (profiles/, PROFILE_FILE) must agree with.  peak 8000 GB/s (MI355X_MICROARCH.md), also as a fraction of the 6290 GB/s measured copy ceiling;
                `bare_launch_ms` = the same call in a loop of bare eager launches (a note, not the claim);
                traffic = the 2*FETCH_SIZE + WRITE_SIZE of the kernel PARSED from that profile's --pmc passes (null when the
                file or the kernel's line is missing -- nothing is hard-coded here).
  elbo_rel_err  |ELBO_hip - ELBO_ref| / |ELBO_ref| on the first 4 096 persons of the benchmark matrix, same parameters
                and noise, in the same run: ref = the CPU restatement of the reference op sequence (fp32, the
                reference's arithmetic); also against its fp64 evaluation.  The sample runs on the kernel the timed step
                runs on (pinned through vibo_desc.flags where the planner would choose differently for 4 096 persons:
                the matrix kernel starts at 4 096 persons / 32 768 at ability_dim <= 4), and the line says so
                (elbo_rel_err_detail.kernel = vibo_plan_kernel's answer).  `also` and `format_p` carry their own.
  roofline.frac_step  the same algorithmic bytes over the whole timed step (ms_per_step), next to the kernel-only frac.
  also_config2  BASELINE configs[1] LITERALLY: 2PL, 100 000 persons x 1 000 items, ability_dim 1, on one GPU -- the same train step
                (a launch of 12.2 batches per workgroup: the kernel's one-shot prologue / end code and the 12-vs-13-round
                quantisation show), with its own kernel and step fractions of the 8 TB/s roofline (N = 1 only).
  extra         config5_path: BASELINE configs[4]'s path (3PL, 10 000 items, conditional, 4 flows) on 100 000 persons;
                other_shapes: configs[3]'s and configs[0]'s matrix shapes (narrow-row kernel), a wide plain matrix, the 3PL link;
                decoder_kernel: the per-term MLP decoder kernel (fwd + bwd, algorithmic TFLOP/s, MFMA issue rate);
                train-step throughput at minibatches of 16 / 4096 / 65536 persons of the resident matrix (SURVEY.md 8d;
                rows gathered in the kernel, hipGraph replay); the headline is the full shard.
  cpu_baseline  the CPU port of the reference op sequence (per-term MLP -> PoE -> link -> masked log-lik -> autograd ->
                Adam, oracle/vibo_oracle.py) timed on this host's cores (rank 0, N = 1 only): B = 16 and B = 1024, 3
                warm-ups + 20 steps each, CPU model and thread count stated; `validated_ratio` = port / REAL reference
                throughput measured in the build container by tools/validate_cpu_port.py (profiles/r02_cpu_port_validation.json).
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
    ap.add_argument('--blocks', type=int, default=5, help='the timed region is run this many times back to back, each exactly --steps steps between barrier + synchronize pairs; value / ms_per_step = the MEDIAN block, all of them listed (ms_per_step_blocks)')
    ap.add_argument('--persons', type=int, default=1_000_000, help='persons of the whole matrix (--scaling strong) / per GPU (--scaling weak)')
    ap.add_argument('--items', type=int, default=1000)
    ap.add_argument('--ability-dim', type=int, default=8)
    ap.add_argument('--also-ability-dim', type=int, default=1, help='second workload reported under "also" (0 = none)')
    ap.add_argument('--irt-model', type=str, default='2pl', choices=['1pl', '2pl', '3pl'])
    ap.add_argument('--missing', type=float, default=0.1)
    ap.add_argument('--lr', type=float, default=5e-3)
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=20)
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='strong', help="'strong' (default): --persons is the whole matrix, split over the ranks (BASELINE configs[2] literally: 1M x 1k over 8 GPUs); 'weak': every rank holds --persons rows")
    ap.add_argument('--no-also-config2', action='store_true', help='skip the BASELINE configs[1] leg (100k x 1k, ability_dim 1: also_config2)')
    ap.add_argument('--no-also-weak', action='store_true', help='N > 1 under strong scaling: skip the extra weak-scaling measurement (also_weak)')
    ap.add_argument('--no-extra', action='store_true', help='skip the minibatch-size sweep and the ELBO rel-err check')
    ap.add_argument('--eval-only', action='store_true', help='forward ELBO only (no backward/optimizer)')
    ap.add_argument('--ability-merge', choices=['product', 'mean'], default='product', help="'mean': the reference's other encoder (models.py:631-650) through VIBO_POSTERIOR_GIVEN + torch autograd / Adam (implies --torch-optimizer --no-graph)")
    ap.add_argument('--no-format-p', action='store_true', help='skip the extra Format P (1-byte cell codes) measurement of the same step')
    ap.add_argument('--torch-optimizer', action='store_true', help='PyTorch autograd + torch.optim.Adam for the O(I) part instead of the fused prologue/epilogue kernels')
    ap.add_argument('--rng', choices=['native', 'torch'], default='native', help='reparameterisation noise: vibo_fill_normal (Philox, in the C ABI) or torch.randn')
    ap.add_argument('--no-graph', action='store_true', help='launch every step eagerly instead of replaying a hipGraph')
    ap.add_argument('--one-graph', action='store_true', help='(default for N > 1; kept for compatibility) the RCCL all-reduce captured inside the step\'s hipGraph: one graph per step, no host round trip')
    ap.add_argument('--two-graphs', action='store_true', help='multi-GPU: two hipGraphs around an EAGER all-reduce instead of the captured collective (both forms are pinned bitwise on a 1-rank nccl group, tests/test_gpu_rccl.py; a failed capture falls back to this form by itself)')
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


def cpu_model_name():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(args, irt):
    """Reference op sequence on the host cores (oracle port), train step, B = 16 and B = 1024."""
    from oracle import vibo_oracle as O
    A, I = args.ability_dim, args.items
    all_threads = torch.get_num_threads()

    def make_step(B):
        g = torch.Generator().manual_seed(args.seed)
        resp, mask = O.simulate_responses(irt, B, I, A, generator=g, missing_frac=0.0)
        mask = mask.long()                      # as the reference's train loop hands it over (vibo.py:240)
        params = {k: v.requires_grad_(True) for k, v in O.init_params(irt, A, I, generator=g).items()}
        opt = torch.optim.Adam(list(params.values()), lr=args.lr)

        def step():
            opt.zero_grad()
            out = O.elbo_forward(params, resp, mask, torch.randn(I, O.item_feat_dim(irt, A)), torch.randn(B, A),
                                 irt_model=irt, ability_dim=A)
            out['loss'].backward()
            opt.step()
        return step

    # the per-term MLP is many mid-sized ops: on a many-core host the default thread count (all cores) is slower than a
    # moderate one, so probe a few and time the best -- the baseline should be the CPU path at its best
    step = make_step(1024)
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
    res = {}
    for B in (16, 1024):
        st = make_step(B)
        for _ in range(3):
            st()
        t0 = time.perf_counter()
        for _ in range(args.cpu_steps):
            st()
        dt = time.perf_counter() - t0
        res[B] = B * I * args.cpu_steps / dt
    torch.set_num_threads(all_threads)
    validated = None
    try:
        v = json.load(open(os.path.join(ROOT, 'profiles', 'r02_cpu_port_validation.json')))
        validated = {'b16': v['b16']['ratio_port_over_reference_throughput'], 'b1024': v['b1024']['ratio_port_over_reference_throughput'],
                     'threads': v['threads'], 'where': 'build container, tools/validate_cpu_port.py (real reference imported there)'}
    except (OSError, KeyError, ValueError):
        pass
    return {
        'value': res[1024], 'unit': 'terms/s', 'cores': best, 'kind': 'port',
        'b16': res[16], 'b1024': res[1024], 'cpu_model': cpu_model_name(), 'host_threads_available': all_threads,
        'validated_ratio': validated,
        'sample': f'3 warm-ups + {args.cpu_steps} train steps each of 16 and of 1024 persons x {I} items (ability_dim {A}, no '
                  f'missing, value = the B = 1024 rate), oracle/vibo_oracle.py = reference op sequence incl. per-term encoder MLP '
                  f'(in-place ELU), torch.distributions Bernoulli, autograd, Adam; thread count = best of a short probe over 8..{all_threads}',
    }


PROFILE_FILE = 'r06_bench_profile.txt'      # rocprofv3 summary of this command on this round's build (tools/collect_profile.sh)


def parse_traffic(kernel_tag):
    """2 * FETCH_SIZE + WRITE_SIZE (KiB counters -> bytes, gfx950 correction of MI355X_MICROARCH.md) of the kernel whose
    mangled name contains `kernel_tag`, from the committed rocprofv3 summary of this command; None if absent."""
    path = os.path.join(ROOT, 'profiles', PROFILE_FILE)
    vals = {}
    try:
        for ln in open(path):
            t = ln.split()
            if len(t) >= 3 and 'msplit_kernel' in t[0] and kernel_tag in t[0] and t[1] in ('FETCH_SIZE', 'WRITE_SIZE'):
                vals.setdefault(t[1], float(t[2]))
    except OSError:
        return None, f'profiles/{PROFILE_FILE} not found'
    if 'FETCH_SIZE' not in vals or 'WRITE_SIZE' not in vals:
        return None, f'no FETCH_SIZE / WRITE_SIZE line for {kernel_tag} in profiles/{PROFILE_FILE}'
    return (2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0, \
        f'parsed from profiles/{PROFILE_FILE}: 2*FETCH_SIZE + WRITE_SIZE of {kernel_tag} (KiB counters, separate --pmc passes)'


def main():
    args = parse()
    if args.ability_merge == 'mean':
        args.torch_optimizer, args.no_graph, args.also_ability_dim = True, True, 0
    irt = int(args.irt_model[0])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # VIBO_BENCH_ONE_DEVICE=1 (test rig, not a measurement): every rank on cuda:0 with the gloo backend -- exercises the N > 1
    # code path (sharding, barriers, max-over-ranks timing, rank-0 JSON) on a box with a single GPU, where RCCL refuses
    # two ranks on one device
    one_device = os.environ.get('VIBO_BENCH_ONE_DEVICE') == '1'
    if one_device:
        local_rank = 0
        args.two_graphs = True          # (gloo's host-side collective cannot be captured into a hipGraph)
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
        if one_device:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from vibo_amd import ops
    from vibo_amd.torch_core.models import VIBO_1PL, VIBO_2PL, VIBO_3PL

    def shard(scaling):
        return args.persons if scaling == 'weak' else (args.persons * (rank + 1)) // world - (args.persons * rank) // world
    persons_rank = shard(args.scaling)

    def measure(A, codes=False, extra=False, persons=None):
        """-> dict(dt, kern_ms, final_loss, graph) for ability_dim A on this rank's shard."""
        P, I = (persons_rank if persons is None else persons), args.items
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

        # in-situ timer: the matrix kernel stamps its own entry / exit (100 MHz chip-wide clock) inside whatever it runs in --
        # the replayed graphs of the timed region included (ops.InsituTimer, vibo_set_insitu_timer); armed before the captures
        # so that the captured launches carry the block's address
        insitu_tm = ops.InsituTimer(dev) if trainer is not None else None
        if insitu_tm is not None:
            insitu_tm.arm()
        # HIP events around the native call, on the stream it is launched on
        events = []
        native = ops._BACKEND['elbo']
        recording = {'on': False}

        last_call = {}

        def timed_native(*a, **k):
            last_call['a'], last_call['k'] = a, k
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
        launch_mode = 'eager'

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

        # The step is captured into hipGraphs and replayed (the eager PyTorch form of the O(I) part is ~80 tiny launches): as ONE
        # graph, or as two (see `forms` below: the faster form of a short probe runs the timed region; config.launch says
        # which).  The fused call's duration inside the step comes from the step's two halves replayed right after the timed
        # region with HIP events between them (see below).
        eager_step = step
        eager_collective = False
        forms, form_probe = None, None
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
                eager_collective = dist is not None and trainer is not None and args.two_graphs
                if not eager_collective:
                    try:
                        with torch.cuda.graph(g):
                            static_loss = step()
                        step_one = lambda: (g.replay(), static_loss)[1]
                        step, launch_mode = step_one, ('one hipGraph per step' if dist is None else 'one hipGraph per step, RCCL all-reduce captured inside')
                        if trainer is not None:
                            # The same step as TWO graphs (fused ELBO call [+ finalize] | [all-reduce +] epilogue) replayed back to back.
                            # Which form is faster depends on the box, reproducibly within a process (same buffers, interleaved, 1M
                            # persons: 0.998 vs 0.892 ms on one box, 0.936 vs 0.940 on another; 125 000 persons: 0.154 vs 0.159):
                            # both are captured, timed for a few steps after the warm-up, and the faster one runs the timed region.
                            ga2, gb2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                            with torch.cuda.graph(ga2, pool=g.pool()):
                                raw2 = trainer.forward_backward(resp, mask)
                            with torch.cuda.graph(gb2, pool=g.pool()):
                                if dist is not None:
                                    dist.all_reduce(raw2.flat)
                                loss2 = trainer.update()
                            step_pair = lambda: (ga2.replay(), gb2.replay(), loss2)[2]
                            forms = {'one': (step_one, launch_mode),
                                     'two': (step_pair, 'two hipGraphs per step (fused ELBO call' + (' + finalize | captured RCCL all-reduce + epilogue)' if dist is not None else ' | epilogue)'))}
                    except Exception as exc:
                        if dist is None or trainer is None:
                            raise
                        print(f'[bench] capturing the collective failed ({type(exc).__name__}: {exc}); eager all-reduce between two graphs', file=sys.stderr)
                        torch.cuda.synchronize()
                        trainer.invalidate()        # (the aborted capture ran forward_backward() on the host: forget the half-open step)
                        eager_collective = True
                        forms = None
                        g = torch.cuda.CUDAGraph()
                if eager_collective:
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        static_raw = trainer.forward_backward(resp, mask)
                    with torch.cuda.graph(g2, pool=g.pool()):
                        static_loss = trainer.update()

                    def step_two():
                        g.replay()
                        dist.all_reduce(static_raw.flat)
                        g2.replay()
                        return static_loss
                    step = step_two
                    launch_mode = 'two hipGraphs per step around an EAGER RCCL all-reduce'
                graph = g
            except Exception as exc:             # never lose the measurement to a capture problem
                print(f'[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); running eagerly', file=sys.stderr)
                graph = None
                step = eager_step
                torch.cuda.synchronize()
                if trainer is not None:
                    trainer.invalidate()

        for _ in range(args.warmup):
            loss = step()
        torch.cuda.synchronize()
        if forms is not None and graph is not None:
            # pick the graph form for the timed region: 2 x 5 steps of each, interleaved (all ranks take rank 0's choice)
            form_probe = {}
            for rep in range(2):
                for name, (fn, _) in forms.items():
                    fn(); torch.cuda.synchronize()
                    tp = time.perf_counter()
                    for _ in range(5):
                        fn()
                    torch.cuda.synchronize()
                    form_probe[name] = min(form_probe.get(name, 1e9), (time.perf_counter() - tp) / 5 * 1e3)
            pick = torch.tensor([0 if form_probe['one'] <= form_probe['two'] else 1], device=dev)
            if dist is not None:
                dist.broadcast(pick, 0)
            step, launch_mode = forms['one' if int(pick) == 0 else 'two']
        # The timed region, `--blocks` times back to back: each block is exactly --steps steps between barrier + synchronize
        # pairs (max over ranks); the line's value / ms_per_step are the MEDIAN block, every block is listed.  (One block of
        # 20 steps is an 18 ms sample: round 5's three clocks disagreed by more than the margin they were quoted with.)
        recording['on'] = graph is None
        block_dt, block_kern = [], []
        for blk in range(max(1, args.blocks)):
            if insitu_tm is not None:
                insitu_tm.reset()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loss = step()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            dtb = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dtb], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dtb = float(t)
            block_dt.append(dtb)
            if insitu_tm is not None:
                block_kern.append(insitu_tm.read())
        recording['on'] = False
        order = sorted(range(len(block_dt)), key=lambda k: block_dt[k])
        med = order[len(order) // 2]
        dt = block_dt[med]
        # the kernel inside the timed steps, by its own clock: mean over all launches of all blocks (+ the median block's own)
        insitu = None
        if block_kern and all(b.get('launches', 0) == args.steps for b in block_kern):
            means = [b['mean_ms'] for b in block_kern]
            insitu = {'mean_ms': sum(means) / len(means), 'median_block_mean_ms': block_kern[med]['mean_ms'],
                      'min_ms': min(b['min_ms'] for b in block_kern), 'max_ms': max(b['max_ms'] for b in block_kern),
                      'launches': sum(b['launches'] for b in block_kern), 'per_block_mean_ms': means,
                      'note': 'the fused kernel timed by itself INSIDE the timed region\'s replayed steps: earliest workgroup entry to latest '
                              'workgroup exit on the chip-wide 100 MHz clock (s_memrealtime), two device-scope atomics per workgroup, no events, '
                              'no tracer (csrc/vibo_device.hpp: insitu_enter / insitu_exit)'}
            if dist is not None:
                ph = torch.tensor([insitu['mean_ms'], insitu['median_block_mean_ms']], device=dev, dtype=torch.float64)
                dist.all_reduce(ph, op=dist.ReduceOp.MAX)
                insitu['mean_ms'], insitu['median_block_mean_ms'] = float(ph[0]), float(ph[1])
                insitu['note'] += '; mean = max over ranks'
        blocks_info = {'blocks': len(block_dt), 'ms_per_step': [b / args.steps * 1e3 for b in block_dt],
                       'median': dt / args.steps * 1e3, 'min': min(block_dt) / args.steps * 1e3, 'max': max(block_dt) / args.steps * 1e3}
        final_loss = float(loss.detach())

        bare_ms, instep, phase_ms = None, None, None
        if graph is not None and trainer is not None:
            # The fused call's duration INSIDE the replayed step.  Events cannot be recorded inside a replayed graph (torch:
            # "external events are disallowed in rocm"), so the same step is captured once more as its two halves --
            # forward_backward() = the fused ELBO call (+ finalize when sharded), update() = [all-reduce +] epilogue -- and the
            # halves are replayed back to back, right after the timed region, with a HIP event on the launch stream before and
            # after the first: the queue never drains and the kernel runs in the regime of the timed steps.  The two event
            # packets and the extra graph boundary cost the step ~10 us (measured: 169 vs 159 vs 154 us per step on a
            # 125 000-person shard), a few of which fall between the events: the figure is an upper bound of the kernel's
            # own duration (+ <= 1 % at 1M persons); the committed rocprofv3 trace has the kernel alone.
            try:
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, pool=graph.pool()):
                    raw_s = trainer.forward_backward(resp, mask)
                with torch.cuda.graph(gb, pool=graph.pool()):
                    if dist is not None and not eager_collective:
                        dist.all_reduce(raw_s.flat)
                    trainer.update()
                n_ev = max(10, min(args.steps, 50))
                evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3 if eager_collective else 2)] for _ in range(n_ev + 3)]
                for ev in evs:
                    ev[0].record(); ga.replay(); ev[1].record()
                    if eager_collective:
                        dist.all_reduce(raw_s.flat); ev[2].record()
                    gb.replay()
                torch.cuda.synchronize()
                evs = evs[3:]
                ks = [e[0].elapsed_time(e[1]) for e in evs]
                rs = [evs[k][-1].elapsed_time(evs[k + 1][0]) for k in range(len(evs) - 1)]
                instep = {'mean_ms': sum(ks) / len(ks), 'min_ms': min(ks), 'max_ms': max(ks), 'replays': len(ks),
                          'epilogue_half_ms': sum(rs) / len(rs),
                          'note': 'HIP events before and after the first of the step\'s two halves (the fused ELBO call), replayed back to '
                                  'back right after the timed region; epilogue_half_ms = from there to the next step\'s first event'}
                if dist is not None:
                    cols = [instep['mean_ms'], instep['epilogue_half_ms']]
                    if eager_collective:
                        cols.append(sum(e[1].elapsed_time(e[2]) for e in evs) / len(evs))
                    ph = torch.tensor(cols, device=dev, dtype=torch.float64)
                    dist.all_reduce(ph, op=dist.ReduceOp.MAX)
                    phase_ms = {'forward_backward_graph': float(ph[0]),
                                'note': 'max over ranks of the mean over the replays of the step captured as two halves (HIP events on the launch '
                                        'stream between them; each event costs a few us of its own); forward_backward_graph = fused ELBO kernel + finalize'}
                    if eager_collective:
                        phase_ms['all_reduce'] = float(ph[2])
                        phase_ms['update_graph'] = float(ph[1])
                        phase_ms['note'] += ', all_reduce = the eager collective, update_graph = epilogue + Adam + next step\'s head (to the next step\'s start)'
                    else:
                        phase_ms['all_reduce_and_update_graph'] = float(ph[1])
                        phase_ms['note'] += ', all_reduce_and_update_graph = captured all-reduce + epilogue + Adam + next step\'s head (to the next step\'s start)'
                    instep['mean_ms'] = float(ph[0])
            except Exception as exc:
                print(f'[bench] in-step kernel timing failed ({type(exc).__name__}: {exc})', file=sys.stderr)
                torch.cuda.synchronize()
                instep = None
        if graph is not None and instep is None:
            # (--torch-optimizer / --eval-only: the native call timed in an eager pass of the same step right after the timed region)
            recording['on'] = True
            for _ in range(min(args.steps, 10)):
                eager_step()
            torch.cuda.synchronize()
            recording['on'] = False
        kern_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, len(events))
        if last_call and dist is None:
            # a note next to the claim: the same call in a loop of bare eager launches (10 back to back between one pair of events)
            reps = 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            native(*last_call['a'], **last_call['k'])
            e0.record()
            for _ in range(reps):
                native(*last_call['a'], **last_call['k'])
            e1.record()
            torch.cuda.synchronize()
            bare_ms = e0.elapsed_time(e1) / reps
            if instep is None:
                kern_ms = max(kern_ms, bare_ms)
        if instep is not None:
            kern_ms = instep['mean_ms']
        events_ms = kern_ms
        if insitu is not None:
            kern_ms = insitu['mean_ms']
        if insitu_tm is not None:
            insitu_tm.disarm()
        bytes_per_term = 5.0 + 12.0 * A / I
        achieved = bytes_per_term * P * I / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        ops._BACKEND['elbo'] = native
        rel, sweep = None, None
        if not args.no_extra and rank == 0 and dist is None:        # (one rank only: these steps would issue collectives of their own)
            rel = elbo_rel_err(model, resp, mask, A)
            if extra and not codes and trainer is not None:
                sweep = batch_sweep(model, resp, mask, A)
        del resp, mask, model, opt, trainer
        torch.cuda.empty_cache()
        return dict(dt=dt, kern_ms=kern_ms, final_loss=final_loss, graph=graph is not None, rel=rel, sweep=sweep,
                    launch=launch_mode if graph is not None else 'eager', phase_ms=phase_ms, bare_ms=bare_ms, instep=instep, form_probe=form_probe,
                    insitu=insitu, events_ms=events_ms, blocks=blocks_info)

    def elbo_rel_err(model, resp, mask, A, n=4096):
        """ELBO of the same parameters, rows and noise: HIP step vs the CPU restatement of the reference (fp32 and fp64).
        4 096 persons on the SAME kernel the timed step runs (`kernel` = vibo_plan_kernel's answer for this call, pinned to the
        timed call's kernel through vibo_desc.flags where the planner would choose differently for the small sample)."""
        from oracle import vibo_oracle as O
        from vibo_amd import _lib
        I = args.items
        n = min(n, resp.shape[0])
        g = torch.Generator(device=dev).manual_seed(args.seed + 7)
        eps_i = torch.randn(I, O.item_feat_dim(irt, A), device=dev, generator=g)
        eps_a = torch.randn(n, A, device=dev, generator=g)
        codes = isinstance(resp, ops.CellCodes)
        if codes:
            hip_rows, hip_mask = resp.rows(slice(0, n)), None
            r, m = hip_rows.unpack()
        else:
            r, m = resp[:n].contiguous(), mask[:n].contiguous()
            hip_rows, hip_mask = r, m
        mcode = _lib.MASK_CODES if codes else _lib.MASK_U8
        kernel_timed = ops.plan_kernel(model.spec, resp.shape[0], I, mcode, not args.eval_only)
        # the sample is smaller than the timed call: pin the row-split kernel the timed call runs (vibo_desc.flags) where the
        # planner would pick the other one for 4 096 persons (ability_dim <= 4: the VALU kernel below 32 768 persons)
        pin = ops.DESC_FLAGS
        if ops.plan_kernel(model.spec, n, I, mcode, False) != kernel_timed:
            pin |= (_lib.FLAG_KERNEL_MATRIX if kernel_timed.startswith('matrix') else
                    _lib.FLAG_KERNEL_VALU if kernel_timed.startswith('VALU') else 0)
        with ops.desc_flags(pin):
            kernel = ops.plan_kernel(model.spec, n, I, mcode, False)
            with torch.no_grad():
                hip = float(model.elbo(*model(hip_rows, hip_mask, eps_item=eps_i, eps_ability=eps_a)))
        out = {}
        for name, dt_ in (('fp32', torch.float32), ('fp64', torch.float64)):
            params = {k: v.detach().cpu().to(dt_) for k, v in model.state_dict().items()}
            with torch.no_grad():
                ref = float(O.elbo_forward(params, r.cpu().to(dt_), m.cpu(), eps_i.cpu().to(dt_), eps_a.cpu().to(dt_),
                                           irt_model=irt, ability_dim=A)['loss'])
            out[name] = abs(hip - ref) / abs(ref)
        return {'persons': n, 'kernel': kernel, 'kernel_of_the_timed_step': kernel_timed, 'same_kernel': kernel == kernel_timed,
                'vs_reference_op_sequence_fp32': out['fp32'], 'vs_same_in_fp64': out['fp64'], 'elbo_hip': hip}

    def batch_sweep(model, resp, mask, A):
        """Train-step rate at the minibatch sizes SURVEY.md §8d names (rows gathered in the kernel, hipGraph replay)."""
        from vibo_amd.trainer import FusedTrainer
        P, I = resp.shape
        res = {}
        for B in (16, 4096, 65536):
            if B >= P:
                continue
            tr = FusedTrainer(model, lr=args.lr, rng='native', seed=args.seed + B)
            rows = torch.randperm(P, device=dev)[:B].contiguous()
            st = lambda: tr.step(resp, mask, row_index=rows)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    st()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                st()
            n_it = max(20, min(1000, int(1e9 // (B * I)) or 20))
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_it):
                gr.replay()
            torch.cuda.synchronize()
            d = (time.perf_counter() - t0) / n_it
            res[str(B)] = {'us_per_step': d * 1e6, 'terms_per_s': B * I / d}
            del tr, gr
        return res

    def decoder_probe():
        """The per-term MLP decoder kernel (--generative-model deep: csrc/vibo_decoder.hip, the path's one dense contraction):
        fwd + bwd of 50 000 persons x the benchmark's items, HIP events on the launch stream."""
        from vibo_amd import decoder as D
        Bd, H = 50_000, 64
        g = torch.Generator(device=dev).manual_seed(args.seed + 77)
        rn = lambda *sh, sc=1.0: torch.randn(*sh, device=dev, generator=g) * sc
        r = (torch.rand(Bd, I, device=dev, generator=g) < 0.5).float()
        mk = (torch.rand(Bd, I, device=dev, generator=g) >= args.missing).view(torch.uint8)
        a = [r, mk, rn(I, H, sc=0.7), rn(Bd, H, sc=0.7), None, None, None, rn(H, H, sc=0.18), rn(H, sc=0.1), rn(H, sc=0.25),
             rn(1, sc=0.1), 0.0, True]
        for _ in range(2):
            D._launch(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            D._launch(*a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        flop = Bd * I * 3 * 2 * H * H
        return {'workload': f'{Bd} persons x {I} items, per-term 64-64-64-1 decoder (deep), forward + backward, fp32-grade (f16 hi/lo MFMA, 3 passes per product)',
                'ms': ms, 'terms_per_s': Bd * I / (ms * 1e-3), 'algorithmic_TFLOPs': flop / (ms * 1e-3) / 1e12,
                'bound': 'mfma + valu', 'mfma_issued_TFLOPs': 3 * flop / (ms * 1e-3) / 1e12, 'mfma_peak_f16_dense_TFLOPs': 2500.0,
                'mfma_issue_frac_of_f16_peak': 3 * flop / (ms * 1e-3) / 1e12 / 2500.0,
                'mfma_util_profile': 'SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles: profiles/r06_decoder_pmc.txt (20.9 % on the shipped kernel; round 2: 17.1 %)'}

    def config5_probe():
        """BASELINE configs[4] LITERALLY where the GPU's memory allows (3PL, 1 000 000 persons x 10 000 items, conditional posterior, 4
        planar flows, fp32 rows: 40 GB of responses + 10 GB of mask bytes resident; 100 000 persons otherwise): one forward +
        backward call = three passes (cond_pre over all ten 1024-item panels, the matrix kernel's one-launch panel mode, the
        table-gradient pass on the matrix pipe)."""
        from vibo_amd import _lib
        from vibo_amd.ops import ElboSpec
        Ic = 10_000
        Pc = 1_000_000 if torch.cuda.mem_get_info(dev)[0] > 120 * 2 ** 30 else 100_000
        g = torch.Generator(device=dev).manual_seed(args.seed + 5)
        r = torch.empty(Pc, Ic, device=dev)
        mk = torch.empty(Pc, Ic, dtype=torch.bool, device=dev)
        for s0 in range(0, Pc, 50_000):          # (in person slices: the temporaries of a one-shot draw would be 90 GB)
            n = min(50_000, Pc - s0)
            r[s0:s0 + n] = (torch.rand(n, Ic, device=dev, generator=g) < 0.5).float()
            mk[s0:s0 + n] = torch.rand(n, Ic, device=dev, generator=g) >= args.missing
        spec = ElboSpec(irt_model=3, ability_dim=1, conditional=True, n_flows=4)
        table = torch.randn(2, Ic, 2, device=dev, generator=g) * 0.5
        item = torch.randn(Ic, 3, device=dev, generator=g)
        eps = torch.randn(Pc, 1, device=dev, generator=g)
        flow = torch.randn(4, 3, device=dev, generator=g) * 0.5
        m8, code = ops.prepare_mask(mk)
        call = lambda: ops._hip_launch_elbo(spec, r, m8, code, None, table, item, eps, flow, _lib.REG_SAMPLED, True, Pc)
        for _ in range(2):
            out = call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        bpt = 5.0 + 12.0 / Ic
        res = {'workload': f'3PL, {Pc} persons x {Ic} items, ability_dim 1, conditional posterior, 4 planar flows, fp32 rows: one forward + backward call'
                           + (' (BASELINE configs[4] at its literal size)' if Pc == 1_000_000 else ' (a tenth of BASELINE configs[4]: not enough free HBM for 1M persons)'),
               'ms': ms, 'ms_per_1e9_terms': ms * 1e9 / (Pc * Ic), 'terms_per_s': Pc * Ic / (ms * 1e-3), 'bytes_per_term': bpt,
               'roofline_frac': bpt * Pc * Ic / (ms * 1e-3) / 8e12, 'loss_is_finite': bool(torch.isfinite(out.scalars).all()),
               'hbm_bytes_per_term_by_construction': 8.0,
               'note': 'cond_pre reads 5 B and writes 1 B of cell codes per term, the matrix kernel and the table-gradient pass (matrix pipe, vibo_cmean.hip) read 1 B each'}
        del r, mk, m8, out
        torch.cuda.empty_cache()
        return res

    def shapes_probe():
        """The other BASELINE shapes as one forward + backward call each (kernel path only, HIP events over 5 calls): configs[3]'s
        CritLangAcq matrix shape (535 598 x 95, 20 % missing, padded 16-byte row strides) and configs[0]'s train split (8 000 x 100)
        on the narrow-row kernel (csrc/vibo_narrow.hip), a wide plain matrix (100 000 x 10 000: all panels in one launch, the rows'
        counts resident with the data; also the first call that counts them) and the 3PL link on the headline shape."""
        from vibo_amd import _lib
        from vibo_amd.ops import ElboSpec
        out = {}
        for name, irt_, Pc, Ic, Ac, miss in (('config4_shape_535598x95', 2, 535_598, 95, 1, 0.2), ('config1_shape_8000x100', 2, 8_000, 100, 1, 0.0),
                                             ('plain_2pl_100000x10000', 2, 100_000, 10_000, 1, args.missing),
                                             ('link_3pl_1Mx1k_ability_dim_8', 3, min(args.persons, 1_000_000), 1000, 8, args.missing)):
            g = torch.Generator(device=dev).manual_seed(args.seed + 7)
            r = (torch.rand(Pc, Ic, device=dev, generator=g) < 0.5).float()
            mk = torch.rand(Pc, Ic, device=dev, generator=g) >= miss
            r, mk = ops.pad_rows(r, mk)
            spec = ElboSpec(irt_model=irt_, ability_dim=Ac)
            table = torch.randn(2, 2 * Ac, device=dev, generator=g) * 0.5
            item = torch.randn(Ic, spec.item_dim, device=dev, generator=g)
            eps = torch.randn(Pc, Ac, device=dev, generator=g)
            r2, m8, code = ops.prepare_rows(r, mk)
            call = lambda: ops._hip_launch_elbo(spec, r2, m8, code, None, table, item, eps, None, _lib.REG_KL, True, Pc)
            counting_ms = None
            if Ic > 1024:
                # rows of more than 1024 items: a matrix the process calls with again is counted once (ops._resident_row_counts:
                # the counts depend on the data alone) -- the timed calls below are those of a resident matrix; here the call that
                # still counts its rows first, for the record
                ops.ROW_COUNT_CACHE = False
                for _ in range(2):
                    call()
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                for _ in range(3):
                    call()
                c1.record()
                torch.cuda.synchronize()
                counting_ms = c0.elapsed_time(c1) / 3
                ops.ROW_COUNT_CACHE = True
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record()
            torch.cuda.synchronize()
            # a second pass with the in-situ timer armed: the fused kernel alone (the hook's own exit atomics -- one line shared by up
            # to 1 020 workgroups of the narrow kernel -- lengthen the call, not the stamped interval: hence its own pass)
            tm = ops.InsituTimer(dev)
            with tm:
                call()
                torch.cuda.synchronize()
                tm.reset()
                for _ in range(5):
                    call()
                torch.cuda.synchronize()
                ks = tm.read()
            ms = e0.elapsed_time(e1) / 5
            bpt = 5.0 + 12.0 * Ac / Ic
            out[name] = {'ms': ms, 'terms_per_s': Pc * Ic / (ms * 1e-3), 'bytes_per_term': bpt,
                         'roofline_frac': bpt * Pc * Ic / (ms * 1e-3) / 8e12, 'kernel': ops.plan_kernel(spec, Pc, Ic, code, True)}
            if counting_ms is not None:
                out[name]['ms_with_the_count_pass'] = counting_ms
                out[name]['note'] = ('ms: the matrix is resident -- its whole-row counts were taken once (vibo_row_counts) and are handed to '
                                     'vibo_elbo_fwd_bwd_counts; ms_with_the_count_pass: a first call on new data (5 B/cell count pass + the panels)')
            if ks.get('launches') == 5:          # the fused kernel alone, by its own stamps (single-launch calls of the matrix / narrow kernels)
                out[name]['kernel_ms_insitu'] = ks['mean_ms']
                out[name]['kernel_roofline_frac'] = bpt * Pc * Ic / (ks['mean_ms'] * 1e-3) / 8e12
            del r, mk, r2, m8
        out['note'] = ('one fused forward + backward call per shape, inputs resident, HIP events over 5 back-to-back calls (kernel + the '
                       'finalize launch + launch gaps: the 8 000 x 100 call is launch-bound); kernel_ms_insitu / kernel_roofline_frac = the fused '
                       'kernel alone by its own entry / exit stamps; fractions of the 8 TB/s roofline on 5 + 12 A / I bytes per term')
        return out

    def conditional_probe():
        """--conditional-posterior on the headline shape (2PL, 1M x 1k): one forward + backward call at ability_dim 1 and 8, on
        fp32 rows and on cell codes (the experts' per-person sums and the table-gradient scatter run as one-hot x table
        contractions on the matrix pipe from 4 096 persons per call: csrc/vibo_cmean.hip; at ability_dim 1 on fp32 rows
        the sums are formed inside the matrix row-split kernel)."""
        from vibo_amd import _lib
        from vibo_amd.ops import ElboSpec
        Pc, Ic = min(args.persons, 1_000_000), args.items
        g = torch.Generator(device=dev).manual_seed(args.seed + 6)
        r = (torch.rand(Pc, Ic, device=dev, generator=g) < 0.5).float()
        mk = torch.rand(Pc, Ic, device=dev, generator=g) >= args.missing
        m8, code = ops.prepare_mask(mk)
        cc = ops.pack_cell_codes(r, mk).codes
        out = {}
        for Ac in (1, 8):
            spec = ElboSpec(irt_model=2, ability_dim=Ac, conditional=True)
            table = torch.randn(2, Ic, 2 * Ac, device=dev, generator=g) * 0.5
            item = torch.randn(Ic, Ac + 1, device=dev, generator=g)
            eps = torch.randn(Pc, Ac, device=dev, generator=g)
            for name, rows in (('fp32_rows', (r, m8, code)), ('cell_codes', (cc, cc, _lib.MASK_CODES))):
                call = lambda: ops._hip_launch_elbo(spec, rows[0], rows[1], rows[2], None, table, item, eps, None, _lib.REG_KL, True, Pc)
                for _ in range(2):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    call()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 3
                bytes_per_term = (1.0 if name == 'cell_codes' else 5.0) + 12.0 * Ac / Ic
                out[f'ability_dim_{Ac}_{name}'] = {'ms': ms, 'terms_per_s': Pc * Ic / (ms * 1e-3),
                                                   'roofline_frac': bytes_per_term * Pc * Ic / (ms * 1e-3) / 8e12}
        out['note'] = ('ability_dim 1 on fp32 rows: the matrix kernel gathers the experts itself while it packs the cells (one 5 B/cell stream; '
                       'VIBO_FLAG_COND_THREE_PASS keeps the separate first pass: 2.2-2.3 ms); every other row: first pass, matrix kernel on the '
                       'emitted cell codes, table-gradient pass')
        out['workload'] = f'2PL, {Pc} persons x {Ic} items, conditional posterior: one forward + backward call'
        return out

    P, I, A = persons_rank, args.items, args.ability_dim
    m = measure(A, extra=not args.no_extra)
    dt, kern_ms, final_loss = m['dt'], m['kern_ms'], m['final_loss']
    bytes_per_term = 5.0 + 12.0 * A / I
    achieved = bytes_per_term * P * I / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    total_persons = float(args.persons) * (world if args.scaling == 'weak' else 1)
    also_weak = None
    if world > 1 and args.scaling == 'strong' and not args.no_also_weak:
        # the same step with --persons rows on EVERY rank (weak scaling), in the same run
        mw = measure(A, persons=args.persons)
        also_weak = {'scaling': 'weak', 'persons_per_rank': args.persons, 'value': float(args.persons) * world * I * args.steps / mw['dt'],
                     'unit': 'terms/s', 'ms_per_step': mw['dt'] / args.steps * 1e3, 'kernel_ms': mw['kern_ms'], 'launch': mw['launch']}
    also = None
    if args.also_ability_dim and args.also_ability_dim != A:
        A2 = args.also_ability_dim
        m2 = measure(A2)
        b2 = 5.0 + 12.0 * A2 / I
        also = {'workload': f'same, ability_dim={A2} (BASELINE configs[1] shape at 1M persons)',
                'value': total_persons * I * args.steps / m2['dt'], 'ms_per_step': m2['dt'] / args.steps * 1e3,
                'ms_per_step_blocks': m2.get('blocks'), 'kernel_ms_events': m2.get('events_ms'),
                'kernel_ms': m2['kern_ms'], 'roofline_achieved_GBps': b2 * P * I / (m2['kern_ms'] * 1e-3) / 1e9,
                'roofline_frac': b2 * P * I / (m2['kern_ms'] * 1e-3) / 1e9 / 8000.0,
                'roofline_frac_step': b2 * total_persons * I * args.steps / m2['dt'] / 1e9 / 8000.0 / world,
                'roofline_frac_of_measured_copy_peak': b2 * P * I / (m2['kern_ms'] * 1e-3) / 1e9 / 6290.0,
                'elbo_rel_err': m2['rel']['vs_reference_op_sequence_fp32'] if m2.get('rel') else None,
                'elbo_rel_err_kernel': m2['rel']['kernel'] if m2.get('rel') else None}

    also_config2 = None
    if world == 1 and not args.no_also_config2 and not args.eval_only:
        # BASELINE configs[1] literally (100k x 1k, ability_dim 1): a short launch, reported with its own fractions
        Pc2 = 100_000
        save = args.no_extra
        args.no_extra = True             # (no rel-err / sweep legs of its own)
        mc = measure(1, persons=Pc2)
        args.no_extra = save
        bc = 5.0 + 12.0 * 1 / I
        also_config2 = {'workload': f'BASELINE configs[1]: {args.irt_model.upper()} simulation, {Pc2} persons x {I} items, ability_dim=1, {args.missing:.0%} missing, one GPU, full-matrix minibatch',
                        'value': float(Pc2) * I * args.steps / mc['dt'], 'unit': 'terms/s', 'ms_per_step': mc['dt'] / args.steps * 1e3,
                        'ms_per_step_blocks': mc.get('blocks'),
                        'kernel_ms': mc['kern_ms'], 'kernel_ms_insitu': mc['insitu']['mean_ms'] if mc.get('insitu') else None,
                        'kernel_ms_events': mc.get('events_ms'), 'kernel_timing_insitu': mc.get('insitu'),
                        'kernel_timing': mc.get('instep'), 'bare_launch_ms': mc.get('bare_ms'),
                        'bytes_per_term': bc, 'launch': mc['launch'],
                        'roofline_achieved_GBps': bc * Pc2 * I / (mc['kern_ms'] * 1e-3) / 1e9,
                        'roofline_frac': bc * Pc2 * I / (mc['kern_ms'] * 1e-3) / 1e9 / 8000.0,
                        'roofline_frac_step': bc * Pc2 * I * args.steps / mc['dt'] / 1e9 / 8000.0,
                        'note': '3 125 batches of 32 rows over 256 workgroups = 12.2 per workgroup (13 rounds) + the kernel\'s one-shot prologue / end code: a short-launch regime, not the streaming one of the headline'}

    format_p = None
    if not args.no_format_p:
        m3 = measure(A, codes=True)
        b3 = 1.0 + 12.0 * A / I
        format_p = {'workload': 'same matrix and step as the headline, rows stored as 1-byte cell codes (Format P, VIBO_MASK_CODES): '
                                'reported separately, its own bytes figure, never mixed with the headline roofline',
                    'value': total_persons * I * args.steps / m3['dt'], 'unit': 'terms/s', 'ms_per_step': m3['dt'] / args.steps * 1e3,
                    'kernel_ms': m3['kern_ms'], 'bytes_per_term': b3,
                    'roofline_achieved_GBps': b3 * P * I / (m3['kern_ms'] * 1e-3) / 1e9,
                    'roofline_frac': b3 * P * I / (m3['kern_ms'] * 1e-3) / 1e9 / 8000.0,
                    'bound': 'valu (latency / issue), not hbm', 'final_loss_per_term': m3['final_loss'] / (total_persons * I),
                    'elbo_rel_err': m3['rel']['vs_reference_op_sequence_fp32'] if m3.get('rel') else None,
                    'elbo_rel_err_kernel': m3['rel']['kernel'] if m3.get('rel') else None}

    # HBM bytes per launch: parsed from the committed rocprofv3 summary of this command (tools/collect_profile.sh); only
    # meaningful for the workload that summary was taken on (the default one)
    default_wl = (args.persons, I, irt, abs(args.missing - 0.1) < 1e-9, world) == (1_000_000, 1000, 2, True, 1)
    # (mangled template arguments <IRT, GRAD, RM = 0 fp32 rows, FLOWS = 0, ...> of vibo::msplit_kernel: the summary keeps the names' tails)
    kernel_tag = f'ILi{irt}ELb{0 if args.eval_only else 1}ELi0ELb0E'
    traffic, traffic_note = (parse_traffic(kernel_tag) if default_wl and A == 8 else (None, 'not measured for this workload'))
    if format_p is not None and default_wl and A == 8:
        format_p['traffic'], _ = parse_traffic(f'ILi{irt}ELb{0 if args.eval_only else 1}ELi2ELb0E')
    if rank == 0:
        terms = total_persons * I * args.steps
        frac_note = ''
        if m.get('insitu'):
            frac_note += ('frac = algorithmic bytes / kernel_ms_insitu: the fused kernel\'s mean duration over every launch of the timed region\'s '
                          'blocks, stamped by the kernel itself (earliest workgroup entry -> latest workgroup exit on the chip-wide 100 MHz clock; '
                          'kernel_timing_insitu).  kernel_ms_events = the estimate of rounds 4-5, kept beside it: ')
        if m.get('instep'):
            frac_note += ('the fused call\'s mean duration in the step captured as its two halves and replayed back to back right after the timed '
                          'region, HIP events on the launch stream around the first (an upper bound: the event packets cost a few us; kernel_timing '
                          'has mean / min / max' + (', mean = max over ranks' if world > 1 else '') + '); ')
        elif not m.get('insitu'):
            frac_note += 'frac = algorithmic bytes / mean duration of the fused call by HIP events on its stream (eager pass after the timed region); '
        frac_note += ('bare_launch_ms = the same call in a loop of bare eager launches (a note, not the claim); frac_step = the same bytes / ms_per_step '
                      f'(median block); the rocprofv3 kernel-trace average of the same command is in profiles/{PROFILE_FILE}')
        line = {
            'metric': 'person x item ELBO terms/sec (train step: fwd + bwd + all-reduce + Adam)'
                      if not args.eval_only else 'person x item ELBO terms/sec (forward ELBO only)',
            'value': terms / dt, 'unit': 'terms/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'ms_per_step_blocks': m.get('blocks'), 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f32 (the three ability-wide contractions of the matrix kernel: 3-pass f16 hi/lo MFMA products, ~22-bit, fp32 accumulate; everything else fp32)',
            'data': 'synthetic',
            'config': {'workload': f'{args.irt_model.upper()} simulation, {P} persons x {I} items per GPU{" (" + str(args.persons) + " in total, strong scaling)" if args.scaling == "strong" else ""}, '
                                   f'ability_dim={A}, {args.missing:.0%} missing, {"product-of-experts" if args.ability_merge == "product" else "mean-merge"} encoder, '
                                   f'unconditional posterior, full-shard minibatch',
                       'global_batch': int(total_persons), 'parallelism': f'person-sharded dp{world}', 'launch': m['launch'],
                       'optimizer': 'torch.optim.Adam (fused)' if (args.torch_optimizer or args.eval_only) else 'fused prologue/epilogue HIP kernels (Adam)',
                       'noise': 'torch.randn' if (args.torch_optimizer or args.eval_only or args.rng == 'torch') else 'Philox4x32-10 drawn in the prologue kernel (vibo_train_prologue_noise = the vibo_fill_normal streams)',
                       'final_loss_per_term': final_loss / (total_persons * I),
                       'persons_per_rank': P, 'phases_ms': m.get('phase_ms'), 'launch_probe_ms': m.get('form_probe')},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': achieved / 8000.0, 'frac_of_measured_copy_peak': achieved / 6290.0,
                         # the same bytes over the whole timed step (kernel with its prologue, [finalize, all-reduce], epilogue + Adam + noise)
                         'frac_step': bytes_per_term * P * I / (dt / args.steps) / 1e9 / 8000.0,
                         'kernel_ms_insitu': m['insitu']['mean_ms'] if m.get('insitu') else None,
                         'kernel_ms_events': m.get('events_ms'),
                         'kernel_timing_insitu': m.get('insitu'),
                         'frac_note': frac_note,
                         'traffic': traffic, 'traffic_note': traffic_note,
                         'kernel': 'vibo::msplit_kernel (its own prologue forms the item sample and the encoder table)' + (' + the finalize helper' if world > 1 else ''),
                         'kernel_ms': kern_ms, 'kernel_timing': m.get('instep'), 'bare_launch_ms': m.get('bare_ms'),
                         'bytes_per_term': bytes_per_term},
        }
        if also_weak is not None:
            line['also_weak'] = also_weak
        if m.get('rel') is not None:
            line['elbo_rel_err'] = m['rel']['vs_reference_op_sequence_fp32']
            line['elbo_rel_err_detail'] = m['rel']
        if m.get('sweep'):
            line['extra'] = {'decoder_kernel': decoder_probe(), 'config5_path': config5_probe(), 'conditional_posterior': conditional_probe(), 'other_shapes': shapes_probe(), 'minibatch_sweep': m['sweep'], 'note': 'train steps on minibatches of the resident matrix, rows gathered in the kernel through row_index, one hipGraph replay per step; the headline is the full shard'}
        if also is not None:
            line['also'] = also
        if also_config2 is not None:
            line['also_config2'] = also_config2
        if format_p is not None:
            line['format_p'] = format_p
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args, irt)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
