"""Layer geometry for the convolutional autoencoder (host-side integer logic).

Same public surface as the reference's model_sizer (src/cae_tools/models/model_sizer.py):
LayerSpec (:16-67), ModelSpec (:70-109), create_model_spec (:112-162), same JSON form
(spec.json / --layer-definitions-path) and same summary text.  Checked row by row against
945 reference-generated specs in tests/golden/model_sizer.json.gz.
"""
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple, Union

KernelSize = Union[int, Tuple[int, int]]


def _conv_out(size: int, kernel: int, stride: int) -> int:
    """output length of an unpadded convolution: floor((size - kernel) / stride) + 1"""
    return (size - kernel) // stride + 1


def _fit_kernel(size: int, kernel: int, stride: int) -> int:
    """smallest kernel >= `kernel` for which a transposed convolution lands exactly on `size`"""
    k = kernel
    while (size - k) % stride:
        k += 1
    return k


@dataclass(eq=False)
class LayerSpec:
    is_input: bool = True
    kernel_size: KernelSize = 3          # int, or (h, w) when the two differ
    stride: int = 2
    input_dimensions: Optional[Sequence[int]] = None   # (chan, y, x)
    output_dimensions: Optional[Sequence[int]] = None  # (chan, y, x)
    output_padding: int = 0

    # accessors the reference's Encoder/Decoder call
    def get_kernel_size(self):
        return self.kernel_size

    def get_stride(self):
        return self.stride

    def get_input_dimensions(self):
        return self.input_dimensions

    def get_output_dimensions(self):
        return self.output_dimensions

    def get_output_padding(self):
        return self.output_padding

    def kernel_hw(self) -> Tuple[int, int]:
        k = self.kernel_size
        return (int(k[0]), int(k[1])) if isinstance(k, (tuple, list)) else (int(k), int(k))

    def __repr__(self):
        head = "Input" if self.is_input else "Output"
        lines = [f"\t{head} Convolutional Layer:",
                 f"\t\tkernel_size={self.kernel_size}  stride={self.stride}"]
        if self.output_padding:
            lines.append(f"\t\toutput_padding=({self.output_padding})")
        lines.append(f"\t\t{self.input_dimensions} => {self.output_dimensions}")
        return "\n".join(lines) + "\n"

    def save(self):
        k = self.kernel_size
        return {"is_input": self.is_input,
                "kernel_size": list(k) if isinstance(k, tuple) else k,
                "stride": self.stride,
                "output_padding": self.output_padding,
                "input_dimensions": list(self.input_dimensions),
                "output_dimensions": list(self.output_dimensions)}

    def load(self, from_obj):
        k = from_obj["kernel_size"]
        self.is_input = from_obj["is_input"]
        self.kernel_size = tuple(k) if isinstance(k, list) else k
        self.stride = from_obj["stride"]
        self.output_padding = from_obj["output_padding"]
        self.input_dimensions = tuple(from_obj["input_dimensions"])
        self.output_dimensions = tuple(from_obj["output_dimensions"])


class ModelSpec:

    def __init__(self, input_layer_specs=None, output_layer_specs=None):
        self.input_layers = list(input_layer_specs) if input_layer_specs is not None else []
        self.output_layers = list(output_layer_specs) if output_layer_specs is not None else []

    def get_input_layers(self):
        return self.input_layers

    def get_output_layers(self):
        return self.output_layers

    def save(self):
        return {"input_layers": [l.save() for l in self.input_layers],
                "output_layers": [l.save() for l in self.output_layers]}

    def load(self, from_obj):
        def _read(items):
            out = []
            for item in items:
                spec = LayerSpec()
                spec.load(item)
                out.append(spec)
            return out
        self.input_layers = _read(from_obj["input_layers"])
        self.output_layers = _read(from_obj["output_layers"])

    def __repr__(self):
        return ("Input Layers:\n" + "".join(map(repr, self.input_layers))
                + "Output Layers:\n" + "".join(map(repr, self.output_layers)))


def create_model_spec(input_size=(7, 7), input_channels=1, output_size=(28, 28), output_channels=1, stride=2,
                      kernel_size=3, limit=3, input_layer_count=None, output_layer_count=None):
    """Encoder: keep halving (unpadded conv, channels x2) while the smaller side stays >= limit,
    always at least one layer.  Decoder: built backwards from the output size with channels x2
    per step, enlarging the kernel per axis until the transposed convolution lands exactly,
    until the map is no larger than the encoder's final map (always at least one layer)."""
    enc = []
    chan = int(input_channels)
    (cur_y, cur_x) = (int(input_size[0]), int(input_size[1]))
    while True:
        (nxt_y, nxt_x) = (_conv_out(cur_y, kernel_size, stride), _conv_out(cur_x, kernel_size, stride))
        if enc:
            capped = input_layer_count is not None and len(enc) >= input_layer_count
            if capped or min(nxt_x, nxt_y) < limit:
                break
        enc.append(LayerSpec(True, kernel_size, stride, (chan, cur_y, cur_x), (chan * 2, nxt_y, nxt_x)))
        chan *= 2
        (cur_y, cur_x) = (nxt_y, nxt_x)
    (floor_y, floor_x) = (cur_y, cur_x)

    dec = []
    chan = int(output_channels)
    (cur_y, cur_x) = (int(output_size[0]), int(output_size[1]))
    while True:
        if dec:
            capped = output_layer_count is not None and len(dec) >= output_layer_count
            if capped or cur_x <= floor_x or cur_y <= floor_y:
                break
        (k_y, k_x) = (_fit_kernel(cur_y, kernel_size, stride), _fit_kernel(cur_x, kernel_size, stride))
        (src_y, src_x) = (_conv_out(cur_y, k_y, stride), _conv_out(cur_x, k_x, stride))
        kernel = k_x if k_x == k_y else (k_y, k_x)
        dec.insert(0, LayerSpec(False, kernel, stride, (chan * 2, src_y, src_x), (chan, cur_y, cur_x)))
        chan *= 2
        (cur_y, cur_x) = (src_y, src_x)
    return ModelSpec(enc, dec)
