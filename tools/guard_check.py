#!/usr/bin/env python3
"""Out-of-bounds write hunt: every buffer the host wrapper hands to the C ABI (outputs, workspace) is carved out of a
larger allocation with 64 KiB of 0xA5 guard bytes on both sides; after each call the guards must be intact.  Each call
is also made twice, with the buffers pre-filled with 0x00 and with 0xFF bytes (NaN): the results must agree (to the
1e-5 the atomically reduced wave-per-person fallback reproduces itself to; bitwise on the row-split paths), i.e.
nothing may read scratch or outputs it has not written in the same call.
   python tools/guard_check.py [--seconds 60] [--seed 0]        (needs the MI355X; test infrastructure)"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from oracle import vibo_oracle as O
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec

GUARD = 65536
FILL = [0]
d = torch.device('cuda:0')
_live = []


class _Torch:
    """`torch` as seen by vibo_amd.ops, with guarded empty()."""

    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def empty(*shape, dtype=torch.float32, device=None):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        pad = (nbytes + 255) // 256 * 256
        raw = torch.full((GUARD + pad + GUARD,), 0xA5, dtype=torch.uint8, device=device)
        raw[GUARD:GUARD + nbytes] = FILL[0]
        _live.append((raw, nbytes))
        return raw[GUARD:GUARD + nbytes].view(dtype).view(*shape)


def check(tag):
    torch.cuda.synchronize()
    bad = []
    for raw, nbytes in _live:
        lo, hi = raw[:GUARD], raw[GUARD + nbytes:]
        if not bool((lo == 0xA5).all()) or not bool((hi == 0xA5).all()):
            nlo, nhi = int((lo != 0xA5).sum()), int((hi != 0xA5).sum())
            first = int((hi != 0xA5).nonzero()[0]) if nhi else -1
            bad.append((nbytes, nlo, nhi, first))
    _live.clear()
    if bad:
        print('GUARD HIT', tag, bad, flush=True)
    return not bad


def same(x, y, exact):
    if exact:
        return torch.equal(x.view(torch.int32), y.view(torch.int32))
    if not (bool(torch.isfinite(x).all()) and bool(torch.isfinite(y).all())):
        return bool((torch.isfinite(x) == torch.isfinite(y)).all()) and same(torch.nan_to_num(x, 0, 0, 0), torch.nan_to_num(y, 0, 0, 0), False)
    return float((x.double() - y.double()).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=60)
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    ops.torch = _Torch()
    t0, n, hits = time.time(), 0, 0
    fixed = [(2, 5, 16, 6000, True, 0), (2, 5, 7, 6000, True, 0), (2, 1, 16, 100, True, 0), (2, 2, 16, 100, False, 2)]
    while time.time() - t0 < a.seconds:
        if n < len(fixed):
            irt, A, B, I, cond, n_flows = fixed[n]
        else:
            irt = rng.choice([1, 2, 2, 3])
            A = rng.choice([1, 2, 3, 4, 5, 8, 11, 16])
            B = rng.choice([1, 3, 7, 8, 9, 16, 17, 64, 130, 1000])
            I = rng.choice([1, 3, 4, 12, 64, 95, 100, 256, 260, 512, 1000, 1024, 1028, 2500, 6000, 10000])
            cond = rng.random() < 0.4
            n_flows = rng.choice([0, 0, 2, 4])
        given = (not cond) and A <= 8 and 4 <= I <= 32767 and rng.random() < 0.25 and n >= len(fixed)      # VIBO_POSTERIOR_GIVEN
        spec = ElboSpec(irt_model=irt, ability_dim=A, n_flows=n_flows, conditional=cond, given=given)
        try:
            spec.check_supported(I)
        except Exception:
            n += 1
            continue
        g = torch.Generator().manual_seed(rng.randrange(1 << 30))
        resp, mask = O.simulate_responses(irt, B + 5, I, A, generator=g, missing_frac=rng.choice([0.0, 0.2]))
        r_, m_ = ops.pad_rows(resp.to(d), mask.bool().to(d))
        use_codes = rng.random() < 0.4 and not (cond and A > 4) and I >= 4 and A <= 8
        if use_codes:
            r_, m_ = ops.pack_cell_codes(r_, m_), None
        r, m, code = ops.prepare_rows(r_, m_)
        rows = torch.randperm(B + 5, generator=g)[:B].to(d) if rng.random() < 0.5 else None
        if rows is None:
            r, m = r[:B], m[:B]
        scale = rng.choice([0.5, 0.5, 2.0, 6.0])
        def nan_fenced(t):      # inputs sit in front of 1 KiB of NaN: reading past their end must not reach a result
            buf = torch.full((t.numel() + 256,), float('nan'), device=d)
            buf[:t.numel()] = t.to(d).reshape(-1)
            return buf[:t.numel()].view(t.shape)
        table = nan_fenced(torch.randn(*spec.table_shape(I, B), generator=g) * 0.5)
        item = nan_fenced(torch.randn(I, spec.item_dim, generator=g) * scale)
        eps = nan_fenced(torch.randn(B, A, generator=g))
        fl = nan_fenced(torch.randn(n_flows, 2 * A + 1, generator=g) * 0.5) if n_flows else None
        reg = _lib.REG_SAMPLED if n_flows else rng.choice([_lib.REG_KL, _lib.REG_SAMPLED])
        cfg = (irt, A, B, I, cond, n_flows, rows is not None, reg, use_codes, given)
        exact = 4 <= I <= 32767 and (not cond or A <= 4) and A <= 8      # row-split paths: partial records + fp64 finalize, no atomics (9..16 dims: wave-per-person kernel)
        for want_grad in (False, True):
            res = []
            for FILL[0] in (0, 0xFF):
                raw = ops._hip_launch_elbo(spec, r, m, code, rows, table, item, eps, fl, reg, want_grad, B)
                hits += not check(('elbo', want_grad) + cfg)
                n_out = raw.flat.numel() if want_grad else _lib.NUM_SCALARS
                if not bool(torch.isfinite(raw.flat[:n_out]).all()) and scale < 3.0 and n_flows == 0:
                    print('NON-FINITE', ('elbo', want_grad, 'scale', scale) + cfg, flush=True)
                    hits += 1
                res.append([raw.flat[:n_out].clone(), raw.ability_mu.clone(), raw.ability_logvar.clone(), raw.ability.clone()])
            for k, (x, y) in enumerate(zip(*res)):
                if not same(x, y, exact):
                    print('INIT-DEPENDENT', ('elbo', want_grad, 'out', k, 'scale', scale) + cfg, flush=True)
                    hits += 1
        res = []
        for FILL[0] in (0, 0xFF):
            if given:
                break
            res.append(ops._hip_encode(spec, r, m, code, rows, table, B))
            hits += not check(('encode',) + cfg)
        for x, y in zip(*res):
            if not same(x, y, exact):
                print('INIT-DEPENDENT', ('encode',) + cfg, flush=True)
                hits += 1
        FILL[0] = 0
        if not cond and not given:
            S = rng.choice([1, 2, 3, 5])
            out = ops._hip_multi_forward(spec, r, m, code, rows, table, torch.stack([item] * S).contiguous(),
                                         torch.stack([eps] * S).contiguous(), fl, _lib.REG_SAMPLED, B)
            hits += not check(('multi', S, out is None) + cfg)
        ops._hip_decode(spec, eps, item)
        hits += not check(('decode',) + cfg)
        # row statistics, repacking and the mean-merge encoder kernels (their outputs / partial records are guarded too)
        cnt = ops._hip_row_counts(r, m, code, rows)
        hits += not check(('row_counts',) + cfg)
        if not use_codes:
            ops.pack_cell_codes(r_, m_)
            hits += not check(('pack_codes',) + cfg)
        if A <= 8:            # (the mean-merge encoder kernels hold 8 ability dims: --ability-merge mean stops there)
            H = rng.choice([4, 16, 64, 100, 256])
            u, v = nan_fenced(torch.randn(H, generator=g)), nan_fenced(torch.randn(H, generator=g))
            w2, b2 = nan_fenced(torch.randn(2 * A, H, generator=g) * 0.3), nan_fenced(torch.randn(2 * A, generator=g))
            cnt = (cnt & ~0xffff) | (cnt & 0xffff).clamp(min=1)          # (an all-missing row is NaN by contract: keep it out)
            post = ops._hip_mean_encoder_fwd(cnt, u, v, w2, b2)
            hits += not check(('mean_fwd', H) + cfg)
            gparts = ops._hip_mean_encoder_bwd(cnt, u, v, w2, nan_fenced(torch.randn(B, 2 * A, generator=g)))
            hits += not check(('mean_bwd', H) + cfg)
            if not (bool(torch.isfinite(post).all()) and all(bool(torch.isfinite(t).all()) for t in gparts)):
                print('NON-FINITE', ('mean_encoder', H) + cfg, flush=True)
                hits += 1
        ops._hip_decode_mean(spec, torch.stack([eps] * 2).contiguous(), torch.stack([item] * 2).contiguous())
        hits += not check(('decode_mean',) + cfg)
        n += 1
    print(f'{n} configurations, {hits} guard hits', flush=True)
    sys.exit(1 if hits else 0)


if __name__ == '__main__':
    main()
