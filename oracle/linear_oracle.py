"""CPU oracle for the LinearModel path  --  TEST INFRASTRUCTURE.  A restatement of src/cae_tools/models/linear.py:19-35
(Flatten -> nn.Linear -> Unflatten) and of the step of linear_model.py:146-153 (MSELoss :241, Adam(lr, weight_decay) :247),
pinned by tests/golden/linear_*.npz, which tests/golden/make_golden_linear.py produced from the reference's own Linear
module (tests/test_linear_cpu.py)."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


class LinearOracle:

    def __init__(self, in_shape, out_shape, state, lr=1e-3, weight_decay=1e-5):
        self.in_shape, self.out_shape = tuple(in_shape), tuple(out_shape)
        self.p = OrderedDict((k, (v.detach().clone() if torch.is_tensor(v) else torch.as_tensor(np.array(v))).requires_grad_(True))
                             for k, v in state.items())
        self.optim = torch.optim.Adam([{"params": list(self.p.values())}], lr=lr, weight_decay=weight_decay)

    def forward(self, x):
        return F.linear(x.flatten(1), self.p["linear.1.weight"], self.p["linear.1.bias"]).view((x.shape[0],) + self.out_shape)

    def eval_forward(self, x):
        with torch.no_grad():
            return self.forward(x)

    def eval_loss(self, x, t):
        with torch.no_grad():
            return float(F.mse_loss(self.forward(x), t))

    def loss_and_grads(self, x, t):
        loss = F.mse_loss(self.forward(x), t)
        self.optim.zero_grad()
        loss.backward()
        return float(loss.detach())

    def train_step(self, x, t):
        loss = self.loss_and_grads(x, t)
        self.optim.step()
        return loss

    def grads(self):
        return OrderedDict((k, v.grad.detach().clone()) for k, v in self.p.items())

    def state(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.p.items())
