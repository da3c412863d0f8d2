"""apply_cae — command line front end with the reference's arguments
(src/cae_tools/cli/apply_cae.py:28-90): load a model folder, score the input file(s) on the GPU,
write inputs + the denormalised prediction variable to a NetCDF file."""
import argparse
import json
import os

import numpy as np

from ..data.arrays import DataArray, open_mfdataset
from ..models.conv_ae_model import ConvAEModel
from ..models.unet import UNET
from ..models.linear_model import LinearModel
from ..models.var_ae_model import VarAEModel


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("data_paths", nargs="+", help="path to netcdf4 file(s) containing data to which model is applied")
    p.add_argument("output_path", help="path to write the netcdf4 file containing input data plus model outputs")
    p.add_argument("--model-folder", help="folder to save the trained model to", required=True)
    p.add_argument("--input-variables", nargs="+", help="name of the input variable(s) in training/test data", required=False)
    p.add_argument("--prediction-variable", help="name of the prediction variable to create in output data",
                   default="model_output")
    p.add_argument("--mask-variable", type=str, help="name of the mask variable", default=None)
    p.add_argument("--gpus", type=int, default=1, help="build-only: shard the cases over this many GPUs of the node")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    from ._launch import maybe_spawn_ranks
    rc = maybe_spawn_ranks("cae_tools_amd.cli.apply_cae", args.gpus, argv)   # before anything touches the GPU
    if rc is not None:
        if rc:
            raise SystemExit(rc)
        return
    from .. import dp as _dp
    _dp.select_device()     # inside a rank: LOCAL_RANK's GPU before the first allocation
    with open(os.path.join(args.model_folder, "parameters.json")) as f:
        parameters = json.loads(f.read())
    kinds = {"ConvAEModel": ConvAEModel, "UNET": UNET, "VarAEModel": VarAEModel, "LinearModel": LinearModel}
    if parameters["type"] not in kinds:
        raise SystemExit(f"cae_tools_amd implements {sorted(kinds)}; model folder holds a {parameters['type']}")
    mt = kinds[parameters["type"]]()
    mt.load(args.model_folder)

    model_names = mt.get_input_variable_names()
    input_variable_names = args.input_variables
    if not input_variable_names:
        if model_names is None:
            raise Exception("Please specify the input variable names using --input-variables")
        input_variable_names = model_names
    elif model_names is not None and input_variable_names != model_names:
        raise Exception(f"input_variables [{','.join(input_variable_names)}] inconsistent with those used to train "
                        f"the model [{','.join(model_names)}]")

    score_ds = open_mfdataset(args.data_paths, concat_dim="box", combine="nested")
    case_dimension = score_ds[input_variable_names[0]].dims[0]
    for var in (model_names or input_variable_names):
        if tuple(score_ds[var].dims) == (case_dimension,):
            (y_dim, x_dim) = (score_ds.dims["y"], score_ds.dims["x"])
            vals = np.asarray(score_ds[var].values)
            score_ds[var] = DataArray(np.broadcast_to(vals[:, None, None, None], (vals.shape[0], 1, y_dim, x_dim)).copy(),
                                      dims=(case_dimension, "channel", "y", "x"))
    print("Applying model for %d cases" % score_ds[case_dimension].shape[0])
    mt.apply(score_ds, input_variable_names, args.prediction_variable, mask_variable_name=args.mask_variable)
    if int(os.environ.get("RANK", "0")) == 0:       # every rank holds all predictions (all-gather); rank 0 writes
        score_ds.to_netcdf(args.output_path)


if __name__ == "__main__":
    main()
