"""A rank of a torch.distributed.run launch must work on ITS GPU from the first allocation on:
DSDataset uploads to torch.cuda.current_device() and HipEngine is created on it, so train() / apply() / evaluate() and the
CLIs select LOCAL_RANK's device BEFORE they build a data set.  No GPU here: torch.cuda is monkeypatched and the order of
the calls is what is asserted.  Also here: --gpus N is refused for the model families without a data-parallel training
path, and the collective overlap calibration is decided from rank-invariant values only."""
import json
import os

import numpy as np
import pytest
import torch


class _Stop(Exception):
    pass


@pytest.fixture
def rank_env(monkeypatch):
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setenv("WORLD_SIZE", "8")
    log = []
    state = {"dev": 0}
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: state["dev"])

    def set_device(d):
        state["dev"] = int(d)
        log.append(("set_device", int(d)))

    monkeypatch.setattr(torch.cuda, "set_device", set_device)
    import torch.distributed as dist
    monkeypatch.setattr(dist, "is_initialized", lambda: False)

    def init_pg(*a, **k):
        log.append(("init_process_group", state["dev"]))
        raise _Stop()

    monkeypatch.setattr(dist, "init_process_group", init_pg)
    return log, state


def test_select_device_picks_the_local_rank(rank_env):
    (log, state) = rank_env
    from cae_tools_amd import dp
    assert dp.select_device() == 3 and state["dev"] == 3 and log == [("set_device", 3)]
    dp.select_device()                                  # already there: no second call
    assert log == [("set_device", 3)]


def test_select_device_is_a_no_op_outside_a_launch(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    called = []
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: called.append(d))
    from cae_tools_amd import dp
    dp.select_device()
    assert not called


@pytest.mark.parametrize("entry", ["train", "apply", "evaluate"])
def test_model_entry_points_select_the_device_before_any_dataset(rank_env, monkeypatch, entry):
    """DSDataset is the first thing that allocates on the GPU: by the time it is constructed (train / apply) or asked for
    its device arrays (evaluate) the rank's device has been selected."""
    (log, state) = rank_env
    from cae_tools_amd.models import conv_ae_model, base_model

    class Probe:
        def __init__(self, *a, **k):
            log.append(("DSDataset", state["dev"]))
            raise _Stop()

    monkeypatch.setattr(conv_ae_model, "DSDataset", Probe)
    monkeypatch.setattr(base_model, "DSDataset", Probe)
    m = conv_ae_model.ConvAEModel()
    m.normalisation_parameters = [{}, {}, 0.0, 1.0]
    ds = {"lowres": type("V", (), {"dims": ("n", "c", "y", "x")})()}
    with pytest.raises(_Stop):
        if entry == "train":
            m.train(["lowres"], "hires", ds, ds)
        elif entry == "apply":
            m.apply(ds, ["lowres"])
        else:
            class DS:
                def set_normalise_output(self, flag):
                    log.append(("dataset_touched", state["dev"]))
                    raise _Stop()
            m.evaluate(DS())
    # whatever came first that could allocate (process-group init with a device id, or the data set) saw device 3
    assert log[0] == ("set_device", 3), log
    assert all(dev == 3 for (what, dev) in log[1:]), log


def test_engine_refuses_tensors_of_another_device():
    """HipEngine.set_dataset / score / encode / decode dereference plain pointers: a tensor on another GPU must raise, not
    fault.  (The check itself needs no GPU: it compares torch devices.)"""
    from cae_tools_amd.engine import HipEngine, CaeError
    eng = HipEngine.__new__(HipEngine)
    eng.device = torch.device("cuda", 3)

    class T:
        device = torch.device("cuda", 0)
    with pytest.raises(CaeError, match="cuda:0.*cuda:3"):
        eng._same_device(T(), "dataset x")
    T.device = torch.device("cuda", 3)
    eng._same_device(T(), "dataset x")


def _cli_args(tmp_path, method, extra=()):
    return ["--train-inputs", "a.nc", "--test-inputs", "b.nc", "--model-folder", str(tmp_path), "--input-variables", "lowres",
            "--output-variable", "hires", "--method", method, "--gpus", "4"] + list(extra)


@pytest.mark.parametrize("method", ["unet", "var", "linear"])
def test_gpus_is_refused_for_models_without_a_data_parallel_path(tmp_path, monkeypatch, method):
    from cae_tools_amd.cli import train_cae, _launch
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(_launch.subprocess, "call", lambda *a, **k: pytest.fail("ranks were spawned"))
    with pytest.raises(SystemExit, match="conv"):
        train_cae.main(_cli_args(tmp_path, method))


def test_gpus_continue_training_checks_the_stored_model_type(tmp_path, monkeypatch):
    from cae_tools_amd.cli import train_cae, _launch
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    spawned = []
    monkeypatch.setattr(_launch.subprocess, "call", lambda cmd, env=None: spawned.append(cmd) or 0)
    with open(os.path.join(tmp_path, "parameters.json"), "w") as f:
        json.dump({"type": "UNET"}, f)
    with pytest.raises(SystemExit, match="UNET"):
        train_cae.main(_cli_args(tmp_path, "conv", ["--continue-training"]))
    assert not spawned
    with open(os.path.join(tmp_path, "parameters.json"), "w") as f:
        json.dump({"type": "ConvAEModel"}, f)
    train_cae.main(_cli_args(tmp_path, "var", ["--continue-training"]))     # the folder decides, not --method
    assert len(spawned) == 1 and "--nproc-per-node=4" in spawned[0]


def test_calibration_decision_is_rank_invariant():
    """dp_calibrate is collective: with a global batch smaller than the world some shard is empty, and then NO rank may
    enter it (the old per-rank `batch > 0` test let the non-empty ranks in alone: a hang)."""
    from cae_tools_amd.dp import DataParallel

    class Dist:
        def get_world_size(self, group=None):
            return 8

        def get_rank(self, group=None):
            return 0

    class Eng:
        dp_world = 8
        calls = 0

        def dp_train_steps(self, *a):
            pass

        def dp_calibrate(self, *a, **k):
            Eng.calls += 1
            return {"overlap": 1.0, "serial": 2.0}

    dp = DataParallel(Eng(), Dist(), sync_bn=False)
    dp._calibrate(0, None, 0, 1, 4)        # global batch 4 < world 8: rank 0 has a row, ranks 4..7 do not
    assert Eng.calls == 0 and dp.calibration == {}
    dp2 = DataParallel(Eng(), Dist(), sync_bn=False)
    dp2._calibrate(0, None, 0, 0, 512)     # an empty LOCAL shard does not keep a rank out when the global batch covers the world
    assert Eng.calls == 1
