"""FusedTrainer (prologue / fused ELBO / epilogue+Adam kernels) must follow the same parameter trajectory as
the PyTorch path (module + autograd + torch.optim.Adam, vibo.py:243-268) under the same noise."""
import copy

import pytest
import torch

from oracle import vibo_oracle as O
from vibo_amd.torch_core.models import VIBO_1PL, VIBO_2PL, VIBO_3PL
from vibo_amd.trainer import FusedTrainer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cls,A,I,B,beta', [(VIBO_2PL, 1, 1000, 300, 1.0), (VIBO_2PL, 8, 200, 130, 0.5),
                                            (VIBO_3PL, 2, 95, 77, 1.0), (VIBO_1PL, 3, 64, 50, 0.7)])
def test_fused_trainer_matches_torch_adam(cls, A, I, B, beta):
    dev = torch.device('cuda:0')
    irt = cls.IRT
    g = torch.Generator().manual_seed(A * 100 + I)
    resp, mask = O.simulate_responses(irt, B, I, A, generator=g, missing_frac=0.15)
    resp, mask = resp.to(dev), mask.bool().to(dev)
    torch.manual_seed(3)
    ref = cls(A, I, ability_merge='product').to(dev)
    fus = copy.deepcopy(ref)
    opt = torch.optim.Adam(ref.parameters(), lr=5e-3)
    trainer = FusedTrainer(fus, lr=5e-3)
    assert list(fus.state_dict().keys()) == list(ref.state_dict().keys())
    for step in range(4):
        torch.manual_seed(100 + step)
        opt.zero_grad()
        loss_ref = ref.elbo_step(resp, mask, annealing_factor=beta)
        loss_ref.backward()
        opt.step()
        torch.manual_seed(100 + step)
        loss_fus = trainer.step(resp, mask, beta=beta)
        assert abs(float(loss_fus) - float(loss_ref.detach())) < 2e-5 * abs(float(loss_ref.detach())), step
    for (k, a), (_, b) in zip(ref.state_dict().items(), fus.state_dict().items()):
        assert (a - b).abs().max() < 2e-5, k
    assert int(trainer.step_count) == 4
