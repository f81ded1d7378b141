#!/usr/bin/env python3
"""The replay fault of the captured MODULE-path step (known limits), isolated: PURE PyTorch, nothing of this library.
A 4-64-64-16 MLP on a few thousand rows, [zero_grad, forward, backward, capturable Adam] captured into a hipGraph; every replay is
compared with the same step run eagerly from the SAME parameters (copied over before each step: nothing accumulates).
    python tools/repro_graph_replay_bias_grad.py [rows ...] [--foreach] [--no-opt] [--lr0] [-v]
Seen on PyTorch 2.10.0+rocm7.0 / HIP 7.0.51831, MI355X: <= 4 096 rows clean over 60 replays; 8 192 rows: from replay 22 on (18 with
foreach Adam; ALSO with lr = 0, i.e. frozen parameters: the trigger is the replay count, not the data) the 64-float gradient of a
BIAS comes back holding other data (|g| 0.77 against the eager 3.2) while every weight gradient stays right; 16 384 rows: from
the first replay.  The weights' GEMM gradients are never affected -- only the 256-byte bias-gradient buffers."""
import sys
import torch

dev = torch.device('cuda:0')
for rows in [int(a) for a in sys.argv[1:] if not a.startswith('-')] or [512, 4096, 8192, 16384]:
    torch.manual_seed(0)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(4, 64), torch.nn.ELU(), torch.nn.Linear(64, 64), torch.nn.ELU(), torch.nn.Linear(64, 16)).to(dev)
    net, twin = mk(), mk()
    twin.load_state_dict(net.state_dict())
    x = torch.randn(rows, 4, device=dev)
    kw = dict(lr=0.0 if '--lr0' in sys.argv else 5e-3, capturable=True, fused='--foreach' not in sys.argv)
    opt, opt2 = torch.optim.Adam(net.parameters(), **kw), torch.optim.Adam(twin.parameters(), **kw)

    def step(m, o):
        o.zero_grad(set_to_none=False)
        out = m(x)                                   # (the table MLP of the conditional encoder: [rows, 4] -> [rows, 16])
        tau = 1.0 / (torch.exp(out[:, 8:]) + 1e-8)   # a product-of-experts flavoured loss: sums over the rows
        loss = ((out[:, :8] * tau).sum(0) / tau.sum(0)).pow(2).sum() + 0.5 * torch.log(tau.sum(0)).sum()
        loss.backward()
        if '--no-opt' not in sys.argv:
            o.step()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step(net, opt); step(twin, opt2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step(net, opt)
    bad = None
    for it in range(60):
        with torch.no_grad():
            for p, q in zip(net.parameters(), twin.parameters()):
                q.copy_(p)
        torch.cuda.synchronize()
        g.replay(); step(twin, opt2)
        torch.cuda.synchronize()
        for (n, p), q in zip(net.named_parameters(), twin.parameters()):
            e = float((p.grad - q.grad).abs().max() / q.grad.abs().max().clamp_min(1e-30))
            if e > 1e-3:
                bad = bad or (it, n, e)
                if '-v' in sys.argv:
                    print(f'  rows {rows} replay {it} {n}: rel {e:.2e}  graph |g|max {float(p.grad.abs().max()):.3e}  eager {float(q.grad.abs().max()):.3e}')
    print(f'rows {rows}: ' + (f'replay {bad[0]} returned a wrong gradient for {bad[1]} (rel {bad[2]:.2e})' if bad else 'clean over 60 replays'))
