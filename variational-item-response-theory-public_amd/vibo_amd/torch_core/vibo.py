"""VIBO training / evaluation CLI on the MI355X-native ELBO engine.

Drop-in for ``python src/torch_core/vibo.py`` of the reference: same flags and
flag interactions (vibo.py:24-125), same output directory name (:127-142), same
artefacts -- ``checkpoint.pth.tar`` / ``model_best.pth.tar`` with keys
``model_state_dict, epoch, args`` enriched post hoc with ``infer_dict``,
``posterior_predict_samples``, ``missing_imputation_accuracy[_mean]``,
``train_logp``, ``test_logp`` (:478-558), ``train_losses.npy``,
``test_losses.npy``, ``train_times.npy`` (negative, as the reference stores them).

What differs is HOW a step runs: the split's response matrix lives in HBM, a
minibatch is a vector of row indices gathered inside the fused HIP kernel, and
forward+backward of the whole [B,I] block is one launch.

    python -m vibo_amd.torch_core.vibo --irt-model 2pl --dataset 2pl_simulation \
        --num-person 10000 --num-item 100 --cuda
    torchrun --nproc-per-node 8 -m vibo_amd.torch_core.vibo ... --cuda     # persons sharded over GPUs
"""
import argparse
import math
import os
import time

import numpy as np
import torch

from .. import config, ops
from ..datasets import artificially_mask_dataset, load_dataset
from ..utils import AverageMeter, save_checkpoint
from .models import VIBO_1PL, VIBO_2PL, VIBO_3PL

MODELS = {'1pl': VIBO_1PL, '2pl': VIBO_2PL, '3pl': VIBO_3PL}


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--irt-model', type=str, default='1pl', choices=['1pl', '2pl', '3pl'])
    p.add_argument('--dataset', type=str, default='1pl_simulation',
                   choices=['1pl_simulation', '2pl_simulation', '3pl_simulation', 'critlangacq',
                            'duolingo', 'wordbank', 'pisa2015_science', 'score_matrix'])
    p.add_argument('--ability-dim', type=int, default=1)
    p.add_argument('--ability-merge', type=str, default='product', choices=['mean', 'product', 'transformer'])
    p.add_argument('--conditional-posterior', action='store_true', default=False)
    p.add_argument('--generative-model', type=str, default='irt', choices=['irt', 'link', 'deep', 'residual'])
    p.add_argument('--response-dist', type=str, default='bernoulli', choices=['gaussian', 'bernoulli'])
    p.add_argument('--drop-missing', action='store_true', default=False)
    p.add_argument('--artificial-missing-perc', type=float, default=0.)
    p.add_argument('--n-norm-flows', type=int, default=0)
    p.add_argument('--no-infer-dict', action='store_true', default=False)
    p.add_argument('--no-marginal', action='store_true', default=False)
    p.add_argument('--no-test', action='store_true', default=False)
    p.add_argument('--no-predictive', action='store_true', default=False)
    p.add_argument('--num-person', type=int, default=1000)
    p.add_argument('--num-item', type=int, default=100)
    p.add_argument('--num-posterior-samples', type=int, default=400)
    p.add_argument('--hidden-dim', type=int, default=64)
    p.add_argument('--max-num-person')
    p.add_argument('--max-num-item')
    p.add_argument('--out-dir', type=str, default=config.OUT_DIR)
    p.add_argument('--lr', type=float, default=5e-3)
    p.add_argument('--batch-size', type=int, default=16, metavar='N')
    p.add_argument('--epochs', type=int, default=100, metavar='N')
    p.add_argument('--max-iters', type=int, default=-1, metavar='N')
    p.add_argument('--num-workers', type=int, default=0)
    p.add_argument('--anneal-kl', action='store_true', default=False)
    p.add_argument('--beta-kl', type=float, default=1.0)
    p.add_argument('--seed', type=int, default=42, metavar='S')
    p.add_argument('--gpu-device', type=int, default=0)
    p.add_argument('--cuda', action='store_true', default=False)
    # additive (not in the reference)
    p.add_argument('--torch-optimizer', action='store_true', default=False,
                   help='use autograd + torch.optim.Adam for the O(I) part instead of the fused HIP trainer kernels')
    p.add_argument('--row-format', choices=['auto', 'f32', 'codes'], default='auto',
                   help="device-resident rows: 'f32' = the reference's fp32 responses + mask bytes (5 B/cell), 'codes' = "
                        "one byte per cell (same results, a fifth of the memory and HBM traffic); 'auto' = codes "
                        "whenever the fused kernels support them (--cuda, 4..32767 items, ability dim <= 4 with "
                        "--conditional-posterior)")
    p.add_argument('--rng', choices=['torch', 'native'], default='torch',
                   help="reparameterisation noise of the fused trainer: torch.randn (the stream torch.manual_seed(--seed) governs) or "
                        "the library's Philox generator drawn inside the prologue kernel (two launches fewer per step)")
    p.add_argument('--graph-module-step', action='store_true', default=False,
                   help='replay the module + autograd + Adam step of the configurations outside the fused trainer (conditional '
                        'posterior, flows, mean merge, MLP decoders) from a hipGraph too; verified for small problems only: on this '
                        'PyTorch-ROCm stack a replayed autograd backward returns wrong bias gradients once a dense layer has a few '
                        'thousand rows (pure-PyTorch repro: tools/repro_graph_replay_bias_grad.py; DESIGN.md 4)')
    p.add_argument('--no-graph', action='store_true', default=False,
                   help='launch every fused train step eagerly instead of replaying a hipGraph (single-GPU runs)')
    p.add_argument('--store-predictive-samples', action='store_true', default=False,
                   help='keep all S posterior-predictive samples [S,P,I,1] in the checkpoint like the '
                        'reference (default: only their mean, [1,P,I,1])')
    return p


def finalize_args(args):
    """Flag interactions of vibo.py:102-125."""
    if args.n_norm_flows > 0:
        args.no_infer_dict = True
        args.no_predictive = True
    if args.artificial_missing_perc > 0:
        args.no_predictive = False
    if config.IS_REAL_WORLD[args.dataset]:
        args.num_person = None
        args.num_item = None
        if args.max_num_person is not None:
            args.max_num_person = int(args.max_num_person)
        if args.max_num_item is not None:
            args.max_num_item = int(args.max_num_item)
    else:
        args.max_num_person = None
        args.max_num_item = None
    return args


def check_supported(args):
    """The flag table is the reference's (drop-in); the choices outside this engine's path fail here, before any data is
    loaded, instead of deep inside the model constructor."""
    problems = []
    if args.ability_merge == 'transformer':
        problems.append("--ability-merge transformer (the reference asserts it away as well, models.py:262)")
    if args.generative_model != 'irt' and args.hidden_dim > 64:
        problems.append(f"--generative-model {args.generative_model} with --hidden-dim > 64 (the per-term decoder kernel covers widths up to 64)")
    if args.ability_dim > 16:
        problems.append("--ability-dim above 16 (1..8 on the row-split kernels, 9..16 on the wave-per-person kernel)")
    if args.ability_dim > 8 and args.ability_merge == 'mean':
        problems.append("--ability-merge mean with --ability-dim above 8 (its caller-supplied posterior needs the row-split kernels)")
    if args.response_dist != 'bernoulli':
        problems.append("--response-dist gaussian (the reference's loader has no *_continuous datasets either)")
    if problems:
        raise SystemExit('not supported by the MI355X engine: ' + '; '.join(problems))


def out_dir_name(args):
    """vibo.py:127-141."""
    return 'VIBO_{}_{}_{}_{}_{}person_{}item_{}maxperson_{}maxitem_{}maskperc_{}ability_{}_{}_seed{}'.format(
        args.irt_model, args.dataset, args.response_dist, args.generative_model, args.num_person, args.num_item,
        args.max_num_person, args.max_num_item, args.artificial_missing_perc, args.ability_dim,
        args.ability_merge, 'conditional_q' if args.conditional_posterior else 'unconditional_q', args.seed)


class ResidentSplit:
    """One dataset split resident on the device: response f32 [P,I] + mask bool [P,I] (the reference's layout, 5 B per
    cell), or with row_format='codes' one byte per cell (ops.CellCodes; mask is None then)."""

    def __init__(self, dataset, device, row_slice=None, row_format='f32'):
        r, m = dataset.matrix()
        if row_slice is not None:
            r, m = r[row_slice], m[row_slice]
        # rows padded to 16 bytes when the item count is not a multiple of 4 (vector loads of the row-split kernel)
        if row_format == 'codes':
            self.response, self.mask = ops.pack_cell_codes(torch.from_numpy(r).to(device), torch.from_numpy(m).to(device)), None
        else:
            self.response, self.mask = ops.pad_rows(torch.from_numpy(r).to(device), torch.from_numpy(m).to(device))
        self.num_person, self.num_item = self.response.shape
        self.device = device

    def rows(self, index):
        """(response, mask) of the persons `index` as their own small matrices."""
        if isinstance(self.response, ops.CellCodes):
            return self.response.rows(index), None
        return self.response[index], self.mask[index]

    def num_batches(self, batch_size, world_rows=None, world=1):
        """Minibatches per epoch.  Person-sharded: the count of the LARGEST shard (ceil(world_rows / world) rows), the same
        number on every rank -- every step carries one collective, so all ranks must take the same number of steps."""
        rows = self.num_person if world_rows is None else -(-world_rows // world)
        return (rows + batch_size - 1) // batch_size

    def batches(self, batch_size, shuffle, generator=None, n_steps=None):
        """Row-index vectors (int64, on device); the kernel gathers the rows itself.  With n_steps (person-sharded runs) the
        shard is cut into exactly that many nearly equal minibatches instead of ceil(rows / batch_size) of them: a rank
        whose shard is one row shorter must not run one step (= one all-reduce) fewer than its peers."""
        if shuffle:
            order = torch.randperm(self.num_person, device=self.device, generator=generator)
        else:
            order = torch.arange(self.num_person, device=self.device)
        if n_steps is not None and n_steps != (self.num_person + batch_size - 1) // batch_size:
            if self.num_person < n_steps:
                raise ValueError(f'{self.num_person} persons on this rank for {n_steps} steps per epoch: raise --batch-size')
            for part in torch.tensor_split(order, n_steps):
                yield part
            return
        for s in range(0, self.num_person, batch_size):
            yield order[s:s + batch_size]


def annealing_factor(args, epoch, batch_idx, n_batches):
    """vibo.py:223-230."""
    if args.anneal_kl:
        return float(batch_idx + epoch * n_batches + 1) / float(args.epochs // 2 * n_batches)
    return args.beta_kl


class GraphedTrainStep:
    """The fused train step over `batch_size` gathered rows as a hipGraph: per minibatch only the row-index vector is
    refreshed and the graph replayed.  At the reference's default batch size (16, vibo.py:81) a step is launch- and
    Python-bound; this takes it from ~0.6 ms to ~0.05 ms.  Full-size minibatches only (the epoch's last, shorter one
    runs eagerly); the first steps run eagerly too (library / allocator warm-up before capture)."""
    WARMUP = 3

    def __init__(self, trainer, data, batch_size):
        self.trainer, self.data, self.batch_size = trainer, data, batch_size
        self.rows = torch.zeros(batch_size, dtype=torch.int64, device=data.device)
        self.graph, self.loss, self.seen = None, None, 0
        self.generation = getattr(trainer, 'generation', 0)

    def __call__(self, rows, beta):
        tr = self.trainer
        if rows.numel() != self.batch_size:
            return tr.step(self.data.response, self.data.mask, beta=beta, row_index=rows)
        self.seen += 1
        if self.graph is None and self.seen <= self.WARMUP:
            return tr.step(self.data.response, self.data.mask, beta=beta, row_index=rows)
        tr.set_beta(beta)
        self.rows.copy_(rows)
        if self.graph is not None and getattr(tr, 'generation', 0) != self.generation:
            self.graph = None                 # a buffer the capture points at was replaced (trainer.generation): capture again
        self.generation = getattr(tr, 'generation', 0)
        if self.graph is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.loss = tr.step(self.data.response, self.data.mask, row_index=self.rows)
            self.graph = g
        self.graph.replay()
        return self.loss


class GraphedModuleStep:
    """The module-path train step (vibo.py:243-268: zero_grad, model.elbo(*model(r, m), beta), backward, Adam) over
    `batch_size` gathered rows as one hipGraph, for the configurations FusedTrainer's three-kernel step does not cover
    (conditional posterior, planar flows, mean merge): ~100 small launches are recorded once and replayed with one host
    call, the row-index vector and the KL weight live in device buffers that are refreshed before each replay.
    `optimizer` must be torch.optim.Adam(capturable=True) (its step counter stays on the device).  Noise comes from the
    default CUDA generator, whose Philox offset torch advances per replay.  Full-size minibatches only."""
    WARMUP = 3

    def __init__(self, model, optimizer, data, batch_size):
        self.model, self.optimizer, self.data, self.batch_size = model, optimizer, data, batch_size
        self.rows = torch.zeros(batch_size, dtype=torch.int64, device=data.device)
        self.beta = torch.ones((), device=data.device)
        self.graph, self.loss, self.seen = None, None, 0
        # the warm-up steps and the capture share one side stream: autograd keeps each parameter's AccumulateGrad node
        # (and the stream it was created on) alive across steps, and a node from the default stream breaks the capture
        self.side = torch.cuda.Stream(device=data.device)

    def _eager(self, rows, beta):
        # Once the graph exists its kernels zero, accumulate into and read the p.grad buffers allocated before the capture:
        # the epoch's last, shorter minibatch runs through here and must not free them (set_to_none=True would hand them
        # back to the allocator, and every later replay would write gradients into memory that may belong to someone else).
        self.optimizer.zero_grad(set_to_none=self.graph is None)
        loss = self.model.elbo_step(self.data.response, self.data.mask, annealing_factor=beta, row_index=rows)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, rows, beta):
        if rows.numel() != self.batch_size:
            return self._eager(rows, beta)
        self.seen += 1
        if self.graph is None and self.seen <= self.WARMUP:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                loss = self._eager(rows, beta)
            torch.cuda.current_stream().wait_stream(self.side)
            return loss
        self.rows.copy_(rows)
        self.beta.fill_(float(beta))
        if self.graph is None:
            self.model._last_ctx = None                      # (drops the last step's autograd graph)
            # gradient buffers are allocated here, outside the capture, and zeroed / accumulated into by captured kernels.
            # (Letting the captured backward allocate them from the graph's pool went wrong on this stack: after a dozen
            # replays the two 64-float bias gradients of the conditional encoder came back holding another tensor's data.)
            for p in self.model.parameters():
                p.grad = torch.zeros_like(p)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.side):
                self.optimizer.zero_grad(set_to_none=False)
                loss = self.model.elbo_step(self.data.response, self.data.mask, annealing_factor=self.beta, row_index=self.rows)
                loss.backward()
                self.optimizer.step()
                self.loss = loss.detach()
            self.model._last_ctx = None
            self.graph = g
        self.graph.replay()
        return self.loss


def train_epoch(model, optimizer, data, args, epoch, batch_size, trainer=None, graphed=None, n_steps=None):
    model.train()
    n_batches = n_steps if n_steps is not None else data.num_batches(batch_size)
    wsum = torch.zeros((), device=data.device)
    count = 0
    for batch_idx, rows in enumerate(data.batches(batch_size, shuffle=True, n_steps=n_steps)):
        beta = annealing_factor(args, epoch, batch_idx, n_batches)
        if graphed is not None:          # the same fused step, replayed from a hipGraph
            loss = graphed(rows, beta)
        elif trainer is not None:        # fused prologue / ELBO / epilogue+Adam kernels
            loss = trainer.step(data.response, data.mask, beta=beta, row_index=rows)
        else:
            optimizer.zero_grad(set_to_none=True)
            loss = model.elbo_step(data.response, data.mask, annealing_factor=beta, row_index=rows)
            loss.backward()
            if getattr(model, 'needs_grad_allreduce', False):      # person-sharded MLP-decoder model: one flat collective
                loss = model.allreduce_grads(loss)
            optimizer.step()
        wsum += loss.detach() * rows.numel()          # AverageMeter weighting (vibo.py:270), one sync per epoch
        count += rows.numel()
    avg = float(wsum) / max(1, count)
    print('====> Train Epoch: {} Loss: {:.4f}'.format(epoch, avg))
    return avg


def test_epoch(model, data, epoch, batch_size, n_steps=None):
    model.eval()
    wsum = torch.zeros((), device=data.device)
    count = 0
    with torch.no_grad():
        for rows in data.batches(batch_size, shuffle=False, n_steps=n_steps):
            loss = model.elbo_step(data.response, data.mask, row_index=rows)
            if getattr(model, 'needs_grad_allreduce', False):
                # person-sharded MLP-decoder model: its loss is shard-local (this rank's persons + 1 / world of the item terms),
                # unlike the fused IRT path, which reduces inside the forward -- one scalar collective, same step count everywhere
                loss = model.allreduce_loss(loss)
            wsum += loss * rows.numel()
            count += rows.numel()
    avg = float(wsum) / max(1, count)
    print('====> Test Epoch: {} Loss: {:.4f}'.format(epoch, avg))
    return avg


def log_marginal_density(model, data, args, batch_size):
    """vibo.py:322-347: batch-level importance-weighted bound, averaged with batch-size weights."""
    model.eval()
    meter = AverageMeter()
    with torch.no_grad():
        for rows in data.batches(batch_size, shuffle=False):
            r, m = data.rows(rows)
            marginal = model.log_marginal(r, m, num_samples=args.num_posterior_samples)
            meter.update(float(torch.mean(marginal)), rows.numel())
    print('====> Marginal: {:.4f}'.format(meter.avg))
    return meter.avg


def infer_dict(model, data, batch_size):
    """vibo.py:420-454."""
    model.eval()
    mus, lvs = [], []
    with torch.no_grad():
        for rows in data.batches(batch_size, shuffle=False):
            _, amu, alv, _, item_mu, item_lv = model.encode(data.response, data.mask, row_index=rows)
            mus.append(amu.cpu())
            lvs.append(alv.cpu())
    return {'ability_mu': torch.cat(mus, 0), 'ability_logvar': torch.cat(lvs, 0),
            'item_feat_mu': item_mu.detach(), 'item_feat_logvar': item_lv.detach()}


def posterior_predictive(model, data, args, batch_size, keep_samples):
    """vibo.py:349-390: S draws of (ability, item) from the posteriors, decoded.  The mean over S
    is accumulated on the device; the [S,P,I,1] stack is only materialised on request."""
    model.eval()
    S = args.num_posterior_samples
    means, stacks = [], []
    with torch.no_grad():
        for rows in data.batches(batch_size, shuffle=False):
            _, amu, alv, _, imu, ilv = model.encode(data.response, data.mask, row_index=rows)
            a_s = amu + torch.exp(0.5 * alv) * torch.randn((S,) + amu.shape, device=amu.device)
            i_s = imu + torch.exp(0.5 * ilv) * torch.randn((S,) + imu.shape, device=imu.device)
            if keep_samples:
                per = [model.decode(a_s[s], i_s[s]).squeeze(2).cpu() for s in range(S)]
                stacks.append(torch.stack(per))
                means.append(stacks[-1].mean(0))
            elif getattr(model, 'generative_model', 'irt') != 'irt':      # per-term MLP decoders: one decoder launch per draw, mean kept on the device
                acc = model.decode(a_s[0], i_s[0]).squeeze(2)
                for s in range(1, S):
                    acc += model.decode(a_s[s], i_s[s]).squeeze(2)
                means.append((acc / S).cpu())
            else:        # one kernel: mean over the S draws, no [S,B,I] intermediate (vibo_decode_mean)
                means.append(ops.decode_probs_mean(model.spec, a_s, i_s).cpu())
    if keep_samples:
        return {'response': torch.cat(stacks, dim=1).unsqueeze(3)}
    return {'response': torch.cat(means, dim=0).unsqueeze(0).unsqueeze(3)}


def posterior_mean_prediction(model, data, batch_size):
    """vibo.py:392-418."""
    model.eval()
    out = []
    with torch.no_grad():
        for rows in data.batches(batch_size, shuffle=False):
            _, amu, _, _, imu, _ = model.encode(data.response, data.mask, row_index=rows)
            out.append(model.decode(amu, imu).cpu())
    return {'response': torch.cat(out, dim=0).unsqueeze(0)}


def imputation_accuracy(inferred, missing_indices, missing_labels):
    """vibo.py:508-548, vectorised: round the inferred probability, compare at the hidden cells."""
    labels = np.asarray(missing_labels)
    if labels.ndim > 1:
        labels = labels.reshape(labels.shape[0], -1)[:, 0]
    idx = torch.as_tensor(np.asarray(missing_indices))
    pred = torch.round(inferred[idx[:, 0], idx[:, 1]]).reshape(-1)
    return float((pred == torch.as_tensor(labels, dtype=pred.dtype)).float().mean())


def main(argv=None):
    args = finalize_args(build_parser().parse_args(argv))
    check_supported(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(args.gpu_device)))
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    from .. import ops
    if args.cuda:
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    elif ops._BACKEND['elbo'] is not ops._hip_launch_elbo:
        device = torch.device('cpu')          # tests/ swapped in the CPU oracle to exercise this host logic
    else:
        raise SystemExit('the MI355X engine has no CPU path: pass --cuda '
                         '(the reference without --cuda is the CPU baseline)')
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.cuda:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)

    args.out_dir = os.path.join(args.out_dir, out_dir_name(args))
    if rank == 0:
        os.makedirs(args.out_dir, exist_ok=True)

    dataset_name = args.dataset if args.response_dist == 'bernoulli' else f'{args.dataset}_continuous'
    kw = dict(num_person=args.num_person, num_item=args.num_item, ability_dim=args.ability_dim,
              max_num_person=args.max_num_person, max_num_item=args.max_num_item)
    train_dataset = load_dataset(dataset_name, train=True, **kw)
    test_dataset = load_dataset(dataset_name, train=False, **kw)
    if args.artificial_missing_perc > 0:
        train_dataset = artificially_mask_dataset(train_dataset, args.artificial_missing_perc)
    num_item = train_dataset.num_item

    # persons are sharded in contiguous blocks over the ranks
    def shard(n):
        return slice(rank * n // world, (rank + 1) * n // world) if world > 1 else None
    row_format = args.row_format
    codes_ok = bool(args.cuda) and 4 <= num_item <= 32767 and args.ability_dim <= 8      # (9..16 ability dims: the wave-per-person kernel reads fp32 rows)
    if row_format == 'auto':
        row_format = 'codes' if codes_ok else 'f32'
    elif row_format == 'codes' and not codes_ok:
        raise SystemExit('--row-format codes needs --cuda, 4..32767 items and --ability-dim <= 8')
    train = ResidentSplit(train_dataset, device, shard(train_dataset.num_person), row_format)
    test = ResidentSplit(test_dataset, device, shard(test_dataset.num_person), row_format)
    local_bs = max(1, args.batch_size // world)
    # steps per epoch: identical on every rank (one all-reduce per step), derived from the largest shard
    n_batches = train.num_batches(local_bs, train_dataset.num_person, world)
    if world > 1:
        # every rank must be able to cut its shard into that many minibatches -- decided from the shard sizes every rank can
        # compute, BEFORE the first collective (a rank that raised alone would leave its peers hanging in their all-reduce)
        for name, n_total, bs_steps in (('train', train_dataset.num_person, n_batches),
                                        ('test', test_dataset.num_person, test.num_batches(local_bs, test_dataset.num_person, world))):
            smallest = min((r + 1) * n_total // world - r * n_total // world for r in range(world))
            if smallest < bs_steps:
                raise SystemExit(f'{name} split: the smallest person shard has {smallest} rows for {bs_steps} steps per epoch over '
                                 f'{world} ranks: raise --batch-size or use fewer ranks')
    n_train_steps = n_batches if world > 1 else None
    n_test_steps = test.num_batches(local_bs, test_dataset.num_person, world) if world > 1 else None
    if args.max_iters != -1:
        args.epochs = int(math.ceil(args.max_iters / float(n_batches)))
        print(f'Found MAX_ITERS={args.max_iters}, setting EPOCHS={args.epochs}')

    model = MODELS[args.irt_model](
        args.ability_dim, num_item, hidden_dim=args.hidden_dim, ability_merge=args.ability_merge,
        conditional_posterior=args.conditional_posterior, generative_model=args.generative_model,
        response_dist=args.response_dist, replace_missing_with_prior=not args.drop_missing,
        n_norm_flows=args.n_norm_flows).to(device)
    if world > 1:
        model.enable_person_sharding(lambda flat: dist.all_reduce(flat), seed=args.seed, rank=rank)
    trainer = None
    plain = args.ability_merge == 'product' and not args.conditional_posterior and args.n_norm_flows == 0
    from ..trainer import fused_trainer_covers
    if (args.cuda and not args.torch_optimizer and fused_trainer_covers(model, args.hidden_dim)
            and (args.hidden_dim <= 256 or not plain)):      # (else: module + torch.optim.Adam)
        # the whole step natively: FusedTrainer's kernels, (conditional posterior / planar flows) FusedCondFlowTrainer's, or
        # (--ability-merge mean, unconditional posterior) FusedMeanTrainer's -- same Adam arithmetic, 2-12 launches
        # per step, no PyTorch autograd inside the replayed graph
        from ..trainer import FusedTrainer
        trainer = FusedTrainer(model, lr=args.lr, rng=args.rng, seed=args.seed, max_batch=local_bs)
    graphed = None
    # The captured module step is opt-in (--graph-module-step): it is 3-4 x faster at small minibatches and follows the eager
    # step exactly in every configuration tests/test_gpu_trainer.py replays 60-150 times, but on this PyTorch / ROCm stack a
    # replayed autograd backward starts returning a wrong 64-float bias gradient after 7-15 replays once a dense PyTorch
    # layer in the step has a few thousand rows (the conditional posterior's 2 x I-row encoder table from 2 048 rows, the
    # `deep` decoder's item network at 1 500) -- DESIGN.md section 4.  Round 6 isolated it: pure PyTorch shows the same fault
    # (tools/repro_graph_replay_bias_grad.py: 8 192 rows wrong from replay 22 on, also with frozen parameters) -- a fault of the
    # PyTorch-ROCm hipGraph stack, not of this library's launches.  The default stays the eager step.
    module_graph = trainer is None and args.cuda and world == 1 and not args.no_graph and args.graph_module_step
    # (capturable: Adam's step counter and bias corrections stay on the device -- required inside a captured graph)
    # (fused: one multi-tensor launch for all parameters instead of ~40 small ones; same update rule)
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr, capturable=bool(module_graph), fused=bool(args.cuda))
    if trainer is not None and world == 1 and not args.no_graph:
        graphed = GraphedTrainStep(trainer, train, local_bs)      # (multi-GPU: eager steps around the all-reduce)
    elif module_graph:
        graphed = GraphedModuleStep(model, optimizer, train, local_bs)      # conditional / flows / mean merge: module path, replayed

    best_loss = np.inf
    train_losses, test_losses, train_times = np.zeros(args.epochs), np.zeros(args.epochs), np.zeros(args.epochs)
    for epoch in range(args.epochs):
        t0 = time.time()
        train_loss = train_epoch(model, optimizer, train, args, epoch, local_bs, trainer, graphed, n_train_steps)
        if args.cuda:
            torch.cuda.synchronize()
        train_losses[epoch] = train_loss
        train_times[epoch] = t0 - time.time()          # negative, like the reference (vibo.py:467)
        if not args.no_test:
            test_loss = test_epoch(model, test, epoch, local_bs, n_test_steps)
            test_losses[epoch] = test_loss
            is_best, best_loss = test_loss < best_loss, min(test_loss, best_loss)
        else:
            is_best, best_loss = train_loss < best_loss, min(train_loss, best_loss)
        if rank == 0:
            save_checkpoint({'model_state_dict': model.state_dict(), 'epoch': epoch, 'args': args},
                            is_best, folder=args.out_dir)
            np.save(os.path.join(args.out_dir, 'train_losses.npy'), train_losses)
            np.save(os.path.join(args.out_dir, 'train_times.npy'), train_times)
            if not args.no_test:
                np.save(os.path.join(args.out_dir, 'test_losses.npy'), test_losses)

    if world > 1:
        dist.barrier()
    # ---- post-hoc enrichment of both checkpoints (vibo.py:490-558).  Person-sharded: the per-person posteriors of `infer_dict` are
    #      computed by every rank on ITS shard and all-gathered in rank order (contiguous shards: = the split's own order) -- the
    #      O(P x I) encode pass is not repeated over the whole split on rank 0.  The sampled enrichments
    #      (posterior predictive, log marginal: per-batch noise streams of the model's generators, batch-level importance weights)
    #      stay on rank 0 over the whole split, where they draw exactly what a one-GPU run draws.
    names = [n for n in ('checkpoint.pth.tar', 'model_best.pth.tar') if os.path.exists(os.path.join(args.out_dir, n))]
    if world > 1:
        flag = torch.tensor([len(names)], device=device)
        dist.broadcast(flag, 0)              # (every rank sees the same files on a shared file system; rank 0's view decides)
        names = ['checkpoint.pth.tar', 'model_best.pth.tar'][:int(flag)]
    sharded_infer = {}
    if world > 1 and not args.no_infer_dict:
        sizes = [(r + 1) * train_dataset.num_person // world - r * train_dataset.num_person // world for r in range(world)]
        for name in names:
            sd = [torch.load(os.path.join(args.out_dir, name), weights_only=False)['model_state_dict']] if rank == 0 else [None]
            dist.broadcast_object_list(sd, 0)
            model.load_state_dict(sd[0])
            saved_reducer, model._reducer = model._reducer, None      # evaluation is local
            part = infer_dict(model, train, local_bs)
            model._reducer = saved_reducer
            whole = {}
            for key in ('ability_mu', 'ability_logvar'):
                t = part[key].to(device)
                buf = torch.zeros(max(sizes), t.shape[1], device=device, dtype=t.dtype)
                buf[:t.shape[0]] = t
                out = [torch.empty_like(buf) for _ in range(world)]
                dist.all_gather(out, buf)
                whole[key] = torch.cat([o[:n] for o, n in zip(out, sizes)], 0).cpu()
            whole['item_feat_mu'], whole['item_feat_logvar'] = part['item_feat_mu'], part['item_feat_logvar']
            sharded_infer[name] = whole
    if rank == 0:
        if world > 1:
            train, test = ResidentSplit(train_dataset, device, None, row_format), ResidentSplit(test_dataset, device, None, row_format)
            local_bs = args.batch_size
        for name in names:
            path = os.path.join(args.out_dir, name)
            ckpt = torch.load(path, weights_only=False)
            model.load_state_dict(ckpt['model_state_dict'])
            saved_reducer, model._reducer = model._reducer, None      # evaluation is local
            if not args.no_infer_dict:
                ckpt['infer_dict'] = sharded_infer[name] if name in sharded_infer else infer_dict(model, train, local_bs)
            if not args.no_predictive:
                pp = posterior_predictive(model, train, args, local_bs, args.store_predictive_samples)
                ckpt['posterior_predict_samples'] = pp
                if args.artificial_missing_perc > 0:
                    acc = imputation_accuracy(pp['response'].mean(0).squeeze(-1), train_dataset.missing_indices,
                                              train_dataset.missing_labels)
                    ckpt['missing_imputation_accuracy'] = acc
                    print(f'Missing Imputation Accuracy from samples: {acc}')
                    pm = posterior_mean_prediction(model, train, local_bs)
                    acc = imputation_accuracy(pm['response'].squeeze(0).squeeze(-1), train_dataset.missing_indices,
                                              train_dataset.missing_labels)
                    ckpt['missing_imputation_accuracy_mean'] = acc
                    print(f'Missing Imputation Accuracy from mean: {acc}')
            if not args.no_marginal:
                ckpt['train_logp'] = log_marginal_density(model, train, args, local_bs)
                if not args.no_test:
                    ckpt['test_logp'] = log_marginal_density(model, test, args, local_bs)
            model._reducer = saved_reducer
            torch.save(ckpt, path)
            print(f'Train time: {np.abs(train_times[:100]).sum()}')
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
