"""model_sizer host logic against the reference-generated sweep (945 argument sets)."""
from helpers import load_sizer_sweep
from cae_tools_amd.models.model_sizer import create_model_spec, ModelSpec, LayerSpec


def test_sweep_matches_reference():
    sweep = load_sizer_sweep()
    assert len(sweep["rows"]) > 900
    for row in sweep["rows"]:
        a = row["args"]
        kwargs = dict(input_size=tuple(a["input_size"]), input_channels=a["input_channels"],
                      output_size=tuple(a["output_size"]), output_channels=a["output_channels"],
                      stride=a["stride"], kernel_size=a["kernel_size"],
                      input_layer_count=a["input_layer_count"], output_layer_count=a["output_layer_count"])
        if "raises" in row["spec"]:
            try:
                create_model_spec(**kwargs)
            except Exception as ex:
                assert type(ex).__name__ == row["spec"]["raises"]
            else:
                raise AssertionError(f"reference raises for {a}")
        else:
            assert create_model_spec(**kwargs).save() == row["spec"], a


def test_repr_and_roundtrip():
    sweep = load_sizer_sweep()
    spec = create_model_spec(input_size=(24, 20), input_channels=1, output_size=(280, 256), output_channels=1)
    assert repr(spec) == sweep["reprs"]["circle2_repr"]
    hs = ModelSpec()
    hs.load(sweep["reprs"]["handspec_roundtrip"])
    assert repr(hs) == sweep["reprs"]["handspec_repr"]
    assert hs.save() == sweep["reprs"]["handspec_roundtrip"]
    l = LayerSpec()
    assert (l.get_kernel_size(), l.get_stride(), l.get_output_padding()) == (3, 2, 0)
