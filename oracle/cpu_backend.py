"""TEST INFRASTRUCTURE ONLY -- stands in for libvibo_hip.so on a machine without a GPU.

``install(ops)`` swaps the three native entry points of vibo_amd.ops for the CPU
analytic restatement (oracle/vibo_table_ref.py) so the *host logic* -- module
surface, state_dict layout, loss/gradient composition, person-sharded
all-reduce, CLI -- can be tested under ``pytest -m "not gpu"``.  Product code
never calls this; only tests/ do.
"""
import torch

from oracle import vibo_table_ref as T


def _gather(t, row_index):
    if t is None:
        return None
    return t[row_index] if row_index is not None else t


def _rows(response, mask, mask_code, row_index):
    """(response fp32, mask uint8) of the call's rows; mask_code 3 = 1-byte cell codes (0 wrong / 1 right / 2 missing)."""
    if mask_code == 3:
        codes = _gather(mask, row_index)
        return (codes == 1).float(), (codes != 2).to(torch.uint8)
    resp = _gather(response, row_index)
    msk = _gather(mask, row_index) if mask is not None else torch.ones_like(resp, dtype=torch.uint8)
    return resp, msk


def _cfg(spec, reg_mode):
    return dict(irt_model=spec.irt_model, ability_dim=spec.ability_dim,
                conditional_posterior=spec.conditional,
                replace_missing_with_prior=not spec.drop_missing,
                mode='kl' if reg_mode == 0 else 'sampled', given_posterior=getattr(spec, 'given', False))


def install(ops):
    from vibo_amd import _lib
    saved = dict(ops._BACKEND)

    def elbo(spec, response, mask, mask_code, row_index, table, item, eps, flow, reg_mode, want_grad, num_person):
        resp, msk = _rows(response, mask, mask_code, row_index)
        A = spec.ability_dim
        flows = None
        if flow is not None:
            flows = [(f[:A], f[A:2 * A], f[2 * A:2 * A + 1]) for f in flow]
        out = T.fused_elbo_ref(table, item, resp, msk, eps, flow_uhat_w_b=flows, want_grad=want_grad,
                               **_cfg(spec, reg_mode))
        n_table, n_item = table.numel(), item.numel()
        n_flow = spec.n_flows * (2 * A + 1)
        flat = torch.zeros(_lib.NUM_SCALARS + 2 * n_table + n_item + 2 * n_flow, dtype=torch.float32)
        flat[_lib.S_LL], flat[_lib.S_REG] = out['ll'], out['reg']
        flat[_lib.S_KL], flat[_lib.S_LOGQ0] = out['kl_ability'], out['logq0']
        flat[_lib.S_LOGP], flat[_lib.S_LADJ] = out['logp'], out['ladj_sum']
        flat[_lib.S_NOBS] = float((msk != 0).sum())
        raw = ops.RawElbo(flat=flat, n_table=n_table, n_item=n_item, n_flow=n_flow,
                          table_shape=tuple(table.shape), ability_mu=out['ability_mu'],
                          ability_logvar=out['ability_logvar'], ability=out['ability'],
                          ability_k=out['ability_k'] if spec.n_flows else None,
                          ability_ladj=out['ladj'] if spec.n_flows else None)
        if want_grad:
            raw.grad_table(0).copy_(out['g_table'][0])
            raw.grad_table(1).copy_(out['g_table'][1])
            raw.grad_item(tuple(item.shape)).copy_(out['g_item'])
            for s in range(2):
                if n_flow:
                    raw.grad_flow(s).copy_(torch.cat([torch.cat(g) for g in out['g_flow'][s]]))
        return raw

    def encode(spec, response, mask, mask_code, row_index, table, num_person):
        resp, msk = _rows(response, mask, mask_code, row_index)
        B, A = resp.shape[0], spec.ability_dim
        dummy_item = torch.zeros(resp.shape[1], spec.item_dim)
        out = T.fused_elbo_ref(table, dummy_item, resp, msk, torch.zeros(B, A), want_grad=False,
                               **_cfg(spec, 0))
        return out['ability_mu'], out['ability_logvar']

    def counts(response, mask, mask_code, row_index):
        resp, msk = _rows(response, mask, mask_code, row_index)
        k = msk != 0
        return (((resp == 1) & k).sum(1).to(torch.int32) << 16) | k.sum(1).to(torch.int32)

    def _mean_w(counts):
        return ((counts >> 16).float() / (counts & 0xffff).float()).unsqueeze(1)

    def mean_fwd(counts, u, v, w2, b2):
        return torch.nn.functional.elu(u + _mean_w(counts) * v) @ w2.t() + b2

    def mean_bwd(counts, u, v, w2, gpost):
        w = _mean_w(counts)
        z = u + w * v
        a = torch.nn.functional.elu(z)
        gz = (gpost @ w2) * torch.where(z > 0, torch.ones_like(z), a + 1.0)
        return gz.sum(0), (w * gz).sum(0), gpost.t() @ a, gpost.sum(0)

    def decode(spec, ability, item):
        from oracle.vibo_oracle import irt_link
        return irt_link(spec.irt_model, ability, item)

    def multi(spec, response, mask, mask_code, row_index, table, items, eps, flow, reg_mode, num_person):
        return None          # "not covered": the module then loops over single forward launches

    def decode_mean(spec, abilities, items):
        from oracle.vibo_oracle import irt_link
        return torch.stack([irt_link(spec.irt_model, abilities[s], items[s]) for s in range(abilities.shape[0])]).mean(0)

    ops._BACKEND.update(elbo=elbo, encode=encode, decode=decode, multi=multi, decode_mean=decode_mean, counts=counts, mean_fwd=mean_fwd,
                        mean_bwd=mean_bwd)

    def restore():
        ops._BACKEND.update(saved)
    return restore
