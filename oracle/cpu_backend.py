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

    def decoder(response, mask, U, V, L, guess, w1, W2, b2, w3, b3, resid, want_grad, want_prob=False):
        """vibo_decoder_fwd_bwd's outputs (one partial record each) from plain fp32 autograd."""
        with torch.enable_grad():
            return _decoder(response, mask, U, V, L, guess, w1, W2, b2, w3, b3, resid, want_grad, want_prob)

    def _decoder(response, mask, U, V, L, guess, w1, W2, b2, w3, b3, resid, want_grad, want_prob):
        F = torch.nn.functional
        names = ('U', 'V', 'L', 'guess', 'w1', 'W2', 'b2', 'w3', 'b3')
        leaves = {k: (v.clone().requires_grad_(True) if v is not None else None)
                  for k, v in zip(names, (U, V, L, guess, w1, W2, b2, w3, b3))}
        z1 = leaves['V'].unsqueeze(1) + (leaves['U'].unsqueeze(0) if U is not None else 0.0)
        if w1 is not None:
            z1 = z1 + leaves['L'].unsqueeze(2) * leaves['w1']
        o = F.elu(F.elu(z1) @ leaves['W2'].t() + leaves['b2']) @ leaves['w3'] + leaves['b3']
        if resid:
            o = o + resid * leaves['L']
        p = torch.sigmoid(o)
        if guess is not None:
            p = leaves['guess'] + (1.0 - leaves['guess']) * p
        pc = p.clamp(1.1920928955078125e-07, 1.0 - 1.1920928955078125e-07)
        ll = torch.where(response > 0.5, pc.log(), torch.log1p(-pc))
        if mask is not None:
            ll = ll * (mask != 0)
        ll = ll.sum()
        out = {'ll_part': ll.detach().reshape(1)}
        if want_prob:
            out['prob'] = p.detach()
        if want_grad:
            live = [k for k in names if leaves[k] is not None and not (k == 'L' and w1 is None and not resid)]
            grads = dict(zip(live, torch.autograd.grad(ll, [leaves[k] for k in live], allow_unused=True)))
            z = lambda k, ref: grads[k] if grads.get(k) is not None else torch.zeros_like(ref)
            out['dW2'] = z('W2', W2).unsqueeze(0)
            dvec = torch.zeros(1, 4, W2.shape[0])
            dvec[0, 0], dvec[0, 1] = z('b2', b2), z('w3', w3)
            if w1 is not None:
                dvec[0, 2] = z('w1', w1)
            dvec[0, 3, 0] = z('b3', b3)[0]
            out['dvec'] = dvec
            out['dV'] = z('V', V).unsqueeze(0)
            if U is not None:
                out['dU'] = z('U', U).unsqueeze(0)
            if L is not None:
                out['dL'] = z('L', L)
            if guess is not None:
                out['dguess'] = z('guess', guess).unsqueeze(0)
        return out

    def flow_stack(z, packed):
        """vibo_flow_stack_forward/backward as plain autograd ops (flows.py:21-41, 58-66)."""
        D = z.shape[1]
        total = 0.0
        for k in range(packed.shape[0]):
            uhat, w, b = packed[k, :D], packed[k, D:2 * D], packed[k, 2 * D]
            t = torch.tanh(z @ w + b)
            z = z + uhat.unsqueeze(0) * t.unsqueeze(1)
            total = total + torch.log(torch.abs(1.0 + (1.0 - t * t) * torch.dot(w, uhat)) + 1e-8)
        return z, total

    def cond_mean_sum(feature, response, mask, row_index):
        """vibo_code_table_sum_forward/backward as two dense products on indicator matrices (autograd)."""
        if isinstance(response, ops.CellCodes):
            codes = _gather(response.codes, row_index)
            r, m = (codes == 1).to(feature.dtype), (codes != 2).to(feature.dtype)
        else:
            r = _gather(ops.prepare_response(response), row_index)
            m = torch.ones_like(r) if mask is None else (_gather(ops.prepare_mask(mask)[0], row_index) != 0).to(r.dtype)
            r = (r == 1).to(m.dtype) * m
        return m @ feature[0] + r @ (feature[1] - feature[0]), m.sum(1)

    ops._BACKEND.update(decoder=decoder, flow_stack=flow_stack, elbo=elbo, encode=encode, decode=decode, multi=multi, decode_mean=decode_mean, counts=counts, mean_fwd=mean_fwd,
                        mean_bwd=mean_bwd, cond_mean_sum=cond_mean_sum)

    def restore():
        ops._BACKEND.update(saved)
    return restore
