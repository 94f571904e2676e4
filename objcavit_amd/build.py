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
SOURCES = ("capi.hip", "linear.hip", "attention.hip", "patch_embed.hip", "bin_head.hip", "depthwise.hip", "conv_igemm.hip", "encoder_nhwc.hip", "upsample.hip", "pointwise_split.hip", "stem.hip", "depthwise_se.hip", "metrics.hip", "mbconv_fused.hip", "pos_sample.hip", "conv_exact.hip", "token_split3.hip", "tap_interp.hip")
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


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    if os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)            # a failed rebuild must not leave a stale library behind
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
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
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    for o in objs:
        os.remove(o)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
