"""`--gpus N` for the CLIs (build-only flag; the reference is single-device: conv_ae_model.py:294-297).

N > 1 re-runs the same command as N ranks under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) as a
CHILD process and returns its exit code.  It must be called before anything in this process touches the GPU (argument
parsing only): a process that has initialised the GPU is never replaced or forked from.  Inside a rank (WORLD_SIZE set by
the launcher) it is a no-op, and the rank's output is kept to rank 0.
"""
import os
import socket
import subprocess
import sys


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_spawn_ranks(module, gpus, argv):
    """None: carry on in this process.  int: the exit code of the multi-rank run this call made."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        if int(os.environ.get("RANK", "0")) != 0:
            sys.stdout = open(os.devnull, "w")     # rank 0 speaks for the job
        return None
    if gpus is None or int(gpus) <= 1:
        return None
    args = list(sys.argv[1:] if argv is None else argv)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(gpus)}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-m", module] + args
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between the ranks' processes here
    return subprocess.call(cmd, env=env)
