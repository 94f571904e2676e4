"""Build libobjcavit_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m objcavit_amd.build [--force]

The library is built IN-TREE at objcavit_amd/lib/libobjcavit_hip.so so that it
travels with the source snapshot to the GPU box (it is git-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get("OCV_CSRC_DIR") or os.path.join(HERE, "csrc")      # env: patched copies for diagnostic builds (tools/build_variant.sh)
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libobjcavit_hip.so")
SOURCES = ("capi.hip", "linear.hip", "attention.hip", "patch_embed.hip", "bin_head.hip", "depthwise.hip", "conv_igemm.hip", "encoder_nhwc.hip", "upsample.hip", "pointwise_split.hip", "stem.hip", "depthwise_se.hip", "metrics.hip", "mbconv_fused.hip", "pos_sample.hip", "conv_exact.hip", "token_split3.hip", "tap_interp.hip", "pointwise_hl.hip", "xattn_h2.hip", "conv_few.hip", "token_h2.hip", "objects_pad.hip", "bin_edges.hip")
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "objcavit_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, out: str = None) -> str:
    """``out``: write the library THERE (diagnostic variants, tools/build_variant.sh / OCV_LIB_OUT in the environment) --
    objects go to a directory of their own next to it and the product library is not touched."""
    out = out or os.environ.get("OCV_LIB_OUT")
    if out:
        return _build_to(os.path.abspath(out), os.path.abspath(out) + ".objs", verbose)
    if not force and not _stale():
        return LIB_PATH
    return _build_to(LIB_PATH, LIB_DIR, verbose)


def _build_to(lib_path: str, obj_dir: str, verbose: bool) -> str:
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    if os.path.exists(lib_path):
        os.remove(lib_path)            # a failed rebuild must not leave a stale library behind
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-c",
               os.path.join(CSRC, src), "-o", obj] + os.environ.get("OCV_EXTRA_HIPCC_FLAGS", "").split()
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose:
            print(out)
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib_path] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    for o in objs:
        os.remove(o)
    if obj_dir != LIB_DIR:
        os.rmdir(obj_dir)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
