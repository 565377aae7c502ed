"""Builds the C-ABI shared library (hipcc, gfx950) in-tree: newsreclib_amd/libnewsreclib_amd.so.

Freshness is decided by CONTENT, not by mtimes (a tree copied by tar / rsync / a snapshot tool may carry any
timestamps): every translation unit under csrc/ is hashed together with the closure of the files it includes; an
object is recompiled when that hash differs from the one recorded next to it, and the hash over all of them is
compiled into the library (`nrl_build_id()`), so a loaded .so can be checked against the sources beside it.
"""
from __future__ import annotations

import glob
import hashlib
import os
import re
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libnewsreclib_amd.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
BUILD_ID_TU = "nrl_build_id.hip"       # the one unit that is handed the build id (so a kernel edit never recompiles the rest)
_INCLUDE = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)


def sources() -> list[str]:
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "*.hip")))


def _closure(path: str, seen: dict[str, bytes]) -> None:
    path = os.path.normpath(path)
    if path in seen:
        return
    with open(path, "rb") as f:
        data = f.read()
    seen[path] = data
    for inc in _INCLUDE.findall(data.decode("utf-8", "replace")):
        cand = os.path.join(os.path.dirname(path), inc)
        if os.path.exists(cand):
            _closure(cand, seen)


def unit_hash(src: str) -> str:
    """sha256 over the unit, every file it (transitively) includes with quotes, and the compile flags."""
    seen: dict[str, bytes] = {}
    _closure(os.path.join(CSRC, src), seen)
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in sorted(seen):
        h.update(os.path.relpath(path, PKG).encode())
        h.update(b"\0")
        h.update(seen[path])
    return h.hexdigest()


def source_hash() -> str:
    """The build id a library built from the present sources carries (nrl_build_id())."""
    h = hashlib.sha256()
    for src in sources():
        if src != BUILD_ID_TU:
            h.update(src.encode())
            h.update(unit_hash(src).encode())
    return h.hexdigest()[:32]


def library_build_id(path: str = LIB) -> str | None:
    """Build id compiled into an existing library, read without loading it (dlopen would pull in the HIP runtime)."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        data = f.read()
    m = re.search(rb"NRL_BUILD_ID=([0-9a-f]{32})", data)
    return m.group(1).decode() if m else None


def _stale() -> bool:
    return library_build_id() != source_hash()


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile what changed for gfx950 and link the shared library; returns its path."""
    want = source_hash()
    if not force and library_build_id() == want:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for src in sources():                     # the stale translation units compile concurrently
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        stamp = obj + ".hash"
        objs.append(obj)
        extra = []
        if src == BUILD_ID_TU:
            uh = want
            extra = [f'-DNRL_BUILD_ID_VALUE="{want}"']
        else:
            uh = unit_hash(src)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == uh:
            continue
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[newsreclib_amd] " + " ".join(cmd), file=sys.stderr)
        if os.path.exists(stamp):
            os.remove(stamp)
        procs.append((cmd, subprocess.Popen(cmd), stamp, uh))
    failed = None
    for cmd, p, stamp, uh in procs:           # wait for ALL compilers before reporting a failure
        if p.wait() != 0:
            failed = failed or subprocess.CalledProcessError(p.returncode, cmd)
        else:
            with open(stamp, "w") as f:
                f.write(uh + "\n")
    if failed is not None:
        raise failed
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print("[newsreclib_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    got = library_build_id()
    if got != want:
        raise RuntimeError(f"linked library carries build id {got}, expected {want}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
