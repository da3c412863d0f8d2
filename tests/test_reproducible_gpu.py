"""Bitwise run-to-run reproducibility of the training step (SURVEY.md §5 / §7 "deterministic reduction mode").

Every cross-workgroup sum of the ConvAE path is a set of fp64 atomic adds whose order varies from run to run.  The addends
are rounded to a fixed power-of-two grid first (kernels_generic.h acc_add), which makes the adds exact and hence independent
of their order; the few cross-wave sums that went through fp32 LDS atomics now have one writer per slot and a fixed fold
order.  So two runs of the same steps from the same state give the same BITS - not "equal to 1e-16" - in every parameter,
moment, running statistic and loss, whether the steps are replayed from a captured graph, launched one by one, or taken
through the data-parallel path."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(batch, steps, graph=True, mode=1, seed=11, dp=None):
    from cae_tools_amd.engine import HipEngine
    from cae_tools_amd.models.model_sizer import create_model_spec
    from cae_tools_amd.models.encoder import Encoder
    from cae_tools_amd.models.decoder import Decoder
    spec = create_model_spec(input_size=(16, 16), input_channels=1, output_size=(256, 256), output_channels=1)
    torch.manual_seed(seed)
    enc = Encoder(spec.get_input_layers(), encoded_space_dim=32, fc_size=128)
    dec = Decoder(spec.get_output_layers(), encoded_space_dim=32, fc_size=128)
    g = torch.Generator().manual_seed(seed + 1)
    n = 3 * batch
    x = torch.rand((n, 1, 16, 16), generator=g).cuda()
    t = torch.rand((n, 1, 256, 256), generator=g).cuda()
    eng = HipEngine(spec, 128, 32, max_batch=batch, graph=graph, specialised=mode)
    eng.load_state(enc.state_dict(), dec.state_dict())
    eng.set_hyper(lr=1e-3, weight_decay=1e-5)
    eng.set_dataset(0, x, t)
    losses = []
    if dp is None:
        for k in range(steps):
            losses.append(eng.train_step(0, None, (k % 3) * batch, batch))
    else:
        from cae_tools_amd.dp import DataParallel
        par = DataParallel(eng, dp, sync_bn=False, overlap=False)
        par.broadcast_parameters(0)
        for k in range(steps):
            slot = par.train_step(0, None, (k % 3) * batch, batch)
            losses.append(eng.dp_read_losses(slot, 1)[0])
    eng.sync()
    return (losses, eng.params.cpu(), eng.exp_avg.cpu(), eng.exp_avg_sq.cpu(), eng.buffers.cpu())


def _same_bits(a, b, what):
    assert a[0] == b[0], (what, "losses", a[0], b[0])
    for (u, v, name) in zip(a[1:], b[1:], ("params", "exp_avg", "exp_avg_sq", "running statistics")):
        assert torch.equal(u, v), f"{what}: {name} differ in {int((u != v).sum())} of {u.numel()} entries"


@pytest.mark.parametrize("batch", [64, 36, 5])
def test_two_runs_of_eight_steps_give_the_same_bits(batch):
    """64: the benchmark batch; 36: the reference's ragged last batch; 5: the short-row-group paths of the fused Linear kernels"""
    a = _run(batch, 8)
    b = _run(batch, 8)
    _same_bits(a, b, f"batch {batch}")


def test_graph_replay_and_plain_launches_give_the_same_bits():
    _same_bits(_run(64, 4, graph=True), _run(64, 4, graph=False), "graph against plain launches")


@pytest.mark.parametrize("mode", [0, 3, 5], ids=["generic-kernels", "lds-staged-backward-everywhere", "gather-forward"])
def test_the_alternative_kernels_are_reproducible_too(mode):
    """cae_set_kernel_mode: 0 the shape-generic kernels, 3 the LDS-staged backward on every eligible layer, 5 the gather forward"""
    _same_bits(_run(16, 3, mode=mode), _run(16, 3, mode=mode), f"kernel mode {mode}")


def test_data_parallel_steps_give_the_same_bits():
    import torch.distributed as dist
    made = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        made = True
    try:
        _same_bits(_run(64, 4, dp=dist), _run(64, 4, dp=dist), "data-parallel path")
    finally:
        if made:
            dist.destroy_process_group()
