"""Builds the C-ABI shared library (hipcc, gfx950) in-tree: newsreclib_amd/libnewsreclib_amd.so."""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libnewsreclib_amd.so")
SOURCES = ["nrl_api.hip", "nrl_kernels.hip", "nrl_attn_mfma.hip", "nrl_sort.hip"]
HEADERS = ["nrl_common.h", "nrl_gemm.h", "nrl_gemm_bf16x3.h", "nrl_gemm_bf16x3_dma.h", "nrl_rowpanel.h", "nrl_gemm_ws.h", "nrl_wgrad_planes.h", "nrl_news_fused.h", "nrl_kernels.h", "nrl_conv.h", "nrl_gru_fused.h", "nrl_api_lstur.inc", "nrl_api_blocks.inc", os.path.join("..", "..", "include", "newsreclib_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source for gfx950 and link the shared library; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for src in SOURCES:                      # the translation units compile concurrently
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[newsreclib_amd] " + " ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print("[newsreclib_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
