"""train_cae — command line front end with the reference's flags (src/cae_tools/cli/train_cae.py:19-53).

--method conv (ConvAEModel, the hot path), unet (UNET), var (VarAEModel: the reference's default method, whose model
source is missing there - this build's own definition, PARITY UNPINNED) and linear (LinearModel) are
implemented; the other method names are accepted by the parser, as in the reference, and rejected with a clear message
(the reference itself constructs no model for vae/unet_res/srcnn_res/resunet_gan: SURVEY.md fact 3).
Build-only flags: --gpus N (data-parallel ConvAE training over N GPUs of this node, one process per GPU) and --local-bn.
"""
import argparse
import json
import os
import time

import numpy as np

from ..data.arrays import DataArray, open_mfdataset
from ..models.conv_ae_model import ConvAEModel
from ..models.unet import UNET
from ..models.linear_model import LinearModel
from ..models.var_ae_model import VarAEModel
from ..models.model_sizer import ModelSpec


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--train-inputs", nargs="+", help="path(s) to netcdf4 file containing training data", required=True)
    p.add_argument("--test-inputs", nargs="+", help="path(s) to netcdf4 file containing test data", required=True)
    p.add_argument("--model-folder", help="folder to save the trained model to", required=True)
    p.add_argument("--continue-training", action="store_true", help="continue training model")
    p.add_argument("--input-variables", nargs="+", help="name of the input variable(s) in training/test data", required=True)
    p.add_argument("--output-variable", help="name of the output variable in training/test data", required=True)
    p.add_argument("--nr-epochs", type=int, help="number of training epochs", default=500)
    p.add_argument("--latent-size", type=int, help="size of the latent space", default=4)
    p.add_argument("--fc-size", type=int, help="size of the fully-connected layers", default=16)
    p.add_argument("--batch-size", type=int, help="number of images to process in one batch", default=10)
    p.add_argument("--learning-rate", type=float, help="the learning rate", default=0.001)
    for (flag, typ, default, text) in [
            ("--lr-step-size", int, 500, "the schedular steps for the learning rate"),
            ("--lr-gamma", float, 0.5, "decay factor of the scheduled learning rate"),
            ("--lambda-mse", float, 1, "controls the strength of the mse loss in vae"),
            ("--lambda-kl", float, 1, "controls the strength of the kl loss in vae"),
            ("--lambda-l1", float, 0.001, "controls the strength of l1 regularization"),
            ("--lambda-pearson", float, 1, "controls the strength of the pearson loss"),
            ("--lambda-ssim", float, 1, "controls the strength of the ssim loss"),
            ("--lambda-additional", float, 1, "controls the strength of additional regularization"),
            ("--weight-decay", float, 1e-5, "weight decay coefficient"),
            ("--dropout-rate", float, 1e-1, "dropout rate")]:
        p.add_argument(flag, type=typ, help=text, default=default)
    p.add_argument("--additional-loss", type=str, default=None, help="additional loss types ('contrastive', 'histogram', 'perceptual')")
    p.add_argument("--scheduler-type", type=str, default=None, help="scheduler type ('StepLR', 'ReduceLROnPlateau', 'ExponentialLR','CosineAnnealingLR')")
    p.add_argument("--method", choices=["conv", "unet", "unet_res", "srcnn_res", "resunet_gan", "var", "vae", "linear"],
                   default="var", help="methods")
    p.add_argument("--layer-definitions-path", help="specify path of a JSON file with layer definitions", default=None)
    p.add_argument("--stride", type=int, help="stride to use in convolutional layers", default=2)
    p.add_argument("--kernel-size", type=int, help="kernel size to use in convolutional layers", default=3)
    p.add_argument("--input-layer-count", type=int, help="number of input convolutional layers", default=None)
    p.add_argument("--output-layer-count", type=int, help="number of output convolutional layers", default=None)
    p.add_argument("--model-id", type=str, help="specify the model id when creating a model", default=None)
    p.add_argument("--database-path", type=str, help="path to a database to store evaluation results", default=None)
    p.add_argument("--chunk-size", type=int, help="chunk size for xarray", default=1000)
    p.add_argument("--include-coasts", help="include coastal areas", default=False)
    p.add_argument("--mask-variable", type=str, help="name of the mask variable", default=None)
    # build-only: the reference trains on one device
    p.add_argument("--gpus", type=int, default=1, help="train data-parallel on this many GPUs of the node (conv method)")
    p.add_argument("--local-bn", action="store_true", help="with --gpus N: per-GPU BatchNorm statistics instead of statistics "
                   "over the global batch (faster; no longer the single-device arithmetic)")
    return p


def broadcast_case_variables(ds, variables, case_dimension):
    """a (case,) variable becomes (case, 1, y, x) by broadcasting (cli/train_cae.py:73-109)"""
    for var in variables:
        if tuple(ds[var].dims) == (case_dimension,):
            print(f"Variable '{var}' does not follow the dimension (box, channel, y, x). Extending dimensions...")
            (y_dim, x_dim) = (ds.dims["y"], ds.dims["x"])
            vals = np.asarray(ds[var].values)
            ds[var] = DataArray(np.broadcast_to(vals[:, None, None, None], (vals.shape[0], 1, y_dim, x_dim)).copy(),
                                dims=(case_dimension, "channel", "y", "x"))


def main(argv=None):
    args = build_parser().parse_args(argv)
    from ._launch import maybe_spawn_ranks
    from .. import dp as _dp
    if max(int(args.gpus or 1), _dp.env_world()[2]) > 1:
        # only ConvAEModel has a data-parallel training path: any other model under N ranks would be trained N times
        # independently, all ranks writing the same model folder and database.  Decided BEFORE ranks are spawned.
        kind = "ConvAEModel" if args.method == "conv" else args.method
        if args.continue_training:
            with open(os.path.join(args.model_folder, "parameters.json")) as f:
                kind = json.loads(f.read()).get("type")
        if kind != "ConvAEModel":
            raise SystemExit(f"--gpus {args.gpus}: data-parallel training is implemented for --method conv (ConvAEModel) only; "
                             f"this run would train a '{kind}' model - use --gpus 1")
    rc = maybe_spawn_ranks("cae_tools_amd.cli.train_cae", args.gpus, argv)   # before anything touches the GPU
    if rc is not None:
        if rc:
            raise SystemExit(rc)
        return
    _dp.select_device()     # inside a rank: LOCAL_RANK's GPU before the first allocation
    train_ds = open_mfdataset(args.train_inputs, concat_dim="box", combine="nested")
    test_ds = open_mfdataset(args.test_inputs, concat_dim="box", combine="nested")
    case_dimension = train_ds[args.output_variable].dims[0]
    print("Training cases: %d, Test cases: %d" % (train_ds[case_dimension].shape[0], test_ds[case_dimension].shape[0]))
    training_paths = ";".join(args.train_inputs)
    test_paths = ";".join(args.test_inputs)
    broadcast_case_variables(train_ds, args.input_variables, case_dimension)
    broadcast_case_variables(test_ds, args.input_variables, case_dimension)

    if args.continue_training:
        with open(os.path.join(args.model_folder, "parameters.json")) as f:
            parameters = json.loads(f.read())
        kinds = {"ConvAEModel": ConvAEModel, "UNET": UNET, "VarAEModel": VarAEModel, "LinearModel": LinearModel}
        if parameters["type"] not in kinds:
            raise SystemExit(f"cae_tools_amd implements {sorted(kinds)}; model folder holds a {parameters['type']}")
        mt = kinds[parameters["type"]]()
        mt.load(args.model_folder)
        mt.nr_epochs = args.nr_epochs
        mt.lr = args.learning_rate
        mt.batch_size = args.batch_size
    else:
        if args.method == "conv":
            mt = ConvAEModel(fc_size=args.fc_size, encoded_dim_size=args.latent_size, nr_epochs=args.nr_epochs,
                             batch_size=args.batch_size, lr=args.learning_rate)
        elif args.method == "unet":     # cli/train_cae.py:131-135
            mt = UNET(fc_size=args.fc_size, encoded_dim_size=args.latent_size, nr_epochs=args.nr_epochs,
                      batch_size=args.batch_size, lr=args.learning_rate, lambda_l1=args.lambda_l1,
                      lambda_pearson=args.lambda_pearson, database_path=args.database_path,
                      weight_decay=args.weight_decay, dropout_rate=args.dropout_rate)
        elif args.method == "var":      # the reference's default method; its model source is missing there
            mt = VarAEModel(fc_size=args.fc_size, encoded_dim_size=args.latent_size, nr_epochs=args.nr_epochs,
                            batch_size=args.batch_size, lr=args.learning_rate, lambda_mse=args.lambda_mse,
                            lambda_kl=args.lambda_kl, lambda_ssim=args.lambda_ssim, weight_decay=args.weight_decay,
                            database_path=args.database_path)
        elif args.method == "linear":   # cli/train_cae.py:137-138
            mt = LinearModel(batch_size=args.batch_size, nr_epochs=args.nr_epochs, lr=args.learning_rate)
        else:
            raise SystemExit(f"--method {args.method}: cae_tools_amd implements 'conv' (ConvAEModel), 'unet' (UNET), 'var' "
                             "(VarAEModel) and 'linear' (LinearModel)")
        if args.model_id:
            mt.set_model_id(args.model_id)
        if args.layer_definitions_path:
            with open(args.layer_definitions_path) as f:
                spec = ModelSpec()
                spec.load(json.loads(f.read()))
                mt.spec = spec

    if hasattr(mt, "sync_bn"):
        mt.sync_bn = not args.local_bn
    start_time = time.time()
    print("Ready for training process")
    mt.train(args.input_variables, args.output_variable, training_ds=train_ds, testing_ds=test_ds,
             model_path=args.model_folder, training_paths=training_paths, testing_paths=test_paths,
             mask_variable_name=args.mask_variable)
    print(f"Time taken to train: {time.time() - start_time:.2f} seconds")


if __name__ == "__main__":
    main()
