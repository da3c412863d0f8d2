"""Host-side parameter containers with the reference's initialisation.

The reference builds torch.nn layers and lets them initialise themselves; "identical seeds"
parity therefore needs the same random draws in the same order from torch's global CPU
generator.  ParamBag registers bare nn.Parameters / buffers under the reference's state_dict
names and replays exactly those draws (torch.nn.modules.conv._ConvNd.reset_parameters and
torch.nn.Linear.reset_parameters: kaiming_uniform_(a=sqrt(5)) on the weight, then
uniform_(-1/sqrt(fan_in), 1/sqrt(fan_in)) on the bias).  No layer here computes anything.
"""
import math

import torch
from torch import nn
from torch.nn import init


class ParamBag(nn.Module):
    """flat holder: register('encoder_cnn.0.weight', tensor) -> state_dict key of the same name"""

    def __init__(self):
        super().__init__()
        self._names = []

    def add_param(self, name, tensor):
        self._names.append(name)
        self.register_parameter(name.replace(".", "__"), nn.Parameter(tensor))

    def add_buffer(self, name, tensor):
        self._names.append(name)
        self.register_buffer(name.replace(".", "__"), tensor)

    def get(self, name):
        return getattr(self, name.replace(".", "__"))

    def names(self):
        return list(self._names)

    # state_dict under the dotted reference names, in registration order
    def state_dict(self, *args, **kwargs):
        from collections import OrderedDict
        out = OrderedDict()
        for n in self._names:
            out[n] = self.get(n).detach()
        return out

    def load_state_dict(self, state_dict, strict=True):
        missing = [n for n in self._names if n not in state_dict]
        extra = [k for k in state_dict if k not in self._names]
        if strict and (missing or extra):
            raise RuntimeError(f"state_dict mismatch: missing {missing}, unexpected {extra}")
        with torch.no_grad():
            for n in self._names:
                if n in state_dict:
                    self.get(n).copy_(torch.as_tensor(state_dict[n]))


def default_layer_init(weight, bias):
    """what nn.Conv2d / nn.ConvTranspose2d / nn.Linear do in reset_parameters()"""
    init.kaiming_uniform_(weight, a=math.sqrt(5))
    (fan_in, _) = init._calculate_fan_in_and_fan_out(weight)
    if fan_in != 0:
        bound = 1 / math.sqrt(fan_in)
        init.uniform_(bias, -bound, bound)


def add_conv(bag, name, shape, n_bias):
    w = torch.empty(shape)
    b = torch.empty(n_bias)
    default_layer_init(w, b)
    bag.add_param(name + ".weight", w)
    bag.add_param(name + ".bias", b)


def add_batchnorm(bag, name, channels):
    """nn.BatchNorm2d defaults: weight 1, bias 0, running_mean 0, running_var 1 (no random draws)"""
    bag.add_param(name + ".weight", torch.ones(channels))
    bag.add_param(name + ".bias", torch.zeros(channels))
    bag.add_buffer(name + ".running_mean", torch.zeros(channels))
    bag.add_buffer(name + ".running_var", torch.ones(channels))
    bag.add_buffer(name + ".num_batches_tracked", torch.tensor(0, dtype=torch.long))
