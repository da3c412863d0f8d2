"""Build libcae_hip.so in-tree with hipcc for gfx950.

    python -m cae_tools_amd.build            # (re)build what is older than its sources
    python -m cae_tools_amd.build --force

Each .hip source is compiled to an object under csrc/_obj/ (only when it or one of its headers changed, the
sources in parallel), then the objects are linked into csrc/libcae_hip.so.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
INC = os.path.join(ROOT, "include")
LIB = os.path.join(CSRC, "libcae_hip.so")
ARCH = "gfx950"
# source -> headers it includes (csrc/ or include/)
SOURCES = {
    "engine.hip": ["kernels_generic.h", "kernels_s2.h", "kernels_last.h", "kernels_rows.h", "kernels_gemm.h", "kernels_igemm.h", "kernels_ctlds.h", "kernels_ctbwd.h", "kernels_head.h", "dp_comm.h", "cae_hip.h", "trunk_api.h"],
    "ctbwd.hip": ["kernels_generic.h", "kernels_gemm.h", "kernels_ctbwd.h"],
    "unet_engine.hip": ["kernels_unet.h", "kernels_unet_mfma.h", "kernels_unet_lin.h", "kernels_unet_thin.h", "kernels_unet_patch.h", "kernels_generic.h", "kernels_gemm.h", "cae_unet.h", "cae_hip.h"],
    "vae_engine.hip": ["kernels_unet.h", "kernels_vae.h", "cae_vae.h", "cae_hip.h", "trunk_api.h"],
    "linear_engine.hip": ["kernels_unet.h", "kernels_unet_mfma.h", "kernels_vae.h", "cae_linear.h", "cae_hip.h"],
}
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-cuda-compat", "-Wno-pass-failed"]
FLAGS += os.environ.get("CAE_HIPCC_FLAGS", "").split()     # tuning experiments: e.g. CAE_HIPCC_FLAGS=-DIG_KCW=32


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libcae_hip.so")


def _path(name):
    p = os.path.join(CSRC, name)
    return p if os.path.exists(p) else os.path.join(INC, name)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def needs_build():
    if any(_stale(_obj(s), [_path(s)] + [_path(h) for h in hs]) for s, hs in SOURCES.items()):
        return True
    return _stale(LIB, [_obj(s) for s in SOURCES])


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src, headers in SOURCES.items():
        if force or _stale(_obj(src), [_path(src)] + [_path(h) for h in headers]):
            cmd = [hipcc] + FLAGS + ["-I" + INC, "-I" + CSRC, "-c", _path(src), "-o", _obj(src)]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in jobs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(failed))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB + ".tmp"] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
