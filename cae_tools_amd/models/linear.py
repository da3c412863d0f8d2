"""Linear — parameter container of the reference's Linear module (src/cae_tools/models/linear.py:19-35): Flatten ->
nn.Linear(C1*y1*x1, C2*y2*x2) -> Unflatten; state_dict keys linear.1.weight / linear.1.bias, PyTorch default init (same seed
=> bit-identical weights).  The arithmetic runs in libcae_hip (include/cae_linear.h)."""
from ._params import ParamBag, add_conv


class Linear(ParamBag):

    def __init__(self, input_shape, output_shape):
        super().__init__()
        (chan1, y1, x1) = input_shape
        (chan2, y2, x2) = output_shape
        self.input_shape, self.output_shape = tuple(input_shape), tuple(output_shape)
        add_conv(self, "linear.1", (chan2 * y2 * x2, chan1 * y1 * x1), chan2 * y2 * x2)

    def forward(self, x):
        raise RuntimeError("Linear.forward on its own is not a product path: use LinearModel.score / apply / train")
