"""Build libcid.so (the C-ABI HIP library, include/cid.h) in-tree for gfx950.

hipcc cross-compiles without a GPU; the resulting ``consistentid_amd/libcid.so``
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
BUILD = HERE / "_build"
LIB = HERE / "libcid.so"
SOURCES = ["gemm.hip", "conv3x3.hip", "linear_h32.hip", "attn.hip", "xattn.hip", "xattn3.hip", "norm.hip", "misc.hip", "f32.hip"]
# sources that exist in experiment builds only, switched on by a define (build_variant)
VARIANT_SOURCES = {}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# per-file flags.  xattn3 / attn: maxima of finite scores and -inf sentinels only -- without NaN semantics hipcc drops the
# v_max_f32 x, x canonicalisation in front of every max (a fifth of the softmax's VALU instructions)
# linear_h32: no SLP vectorisation -- packed fp32 add / mul are an anti-lever beside MFMAs on gfx950 and the
# epilogue of that kernel runs in the MFMAs' shadow
FILE_FLAGS = {"xattn3.hip": ["-fno-honor-nans"], "attn.hip": ["-fno-honor-nans"], "linear_h32.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def build_variant(name: str, defines, verbose: bool = True) -> Path:
    """Experiment builds: libcid_<name>.so with extra -D flags (selected at run time with CID_LIBRARY=<path>).  An entry that
    starts with '-' is passed to hipcc as it is (e.g. -fno-slp-vectorize): code-generation experiments on unchanged sources;
    "<file>.hip:<flag>" applies the flag (or define) to that one source only."""
    hipcc = _hipcc()
    bdir = BUILD / name
    bdir.mkdir(parents=True, exist_ok=True)
    lib = HERE / f"libcid_{name}.so"
    as_flag = lambda d: d if d.startswith("-") else f"-D{d}"
    per_file = {}
    for d in [d for d in defines if ".hip:" in d]:
        f, flag = d.split(":", 1)
        per_file.setdefault(f, []).append(as_flag(flag))
    defines = [d for d in defines if ".hip:" not in d]
    extra = [as_flag(d) for d in defines]
    sources = SOURCES + [s for d in defines for s in VARIANT_SOURCES.get(d.split("=")[0], [])]

    def compile_one(src: str):
        obj = bdir / (src.replace(".hip", ".o"))
        r = subprocess.run([hipcc, *FLAGS, *FILE_FLAGS.get(src, []), *extra, *per_file.get(src, []), "-c", str(CSRC / src), "-o", str(obj)],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, sources))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(lib)],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print(f"[consistentid_amd] built {lib}", file=sys.stderr)
    return lib


def build(force: bool = False, verbose: bool = True) -> Path:
    BUILD.mkdir(exist_ok=True)
    deps = [CSRC / s for s in SOURCES] + [CSRC / "common.h", CSRC / "gemm_args.h", CSRC / "xattn_frag.h", CSRC / "xattn_core.h", HERE.parent / "include" / "cid.h"]
    stamp = BUILD / "stamp"
    want = _digest(deps)
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == want:
        return LIB
    hipcc = _hipcc()

    def compile_one(src: str):
        obj = BUILD / (src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src, []), "-c", str(CSRC / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    stamp.write_text(want)
    if verbose:
        print(f"[consistentid_amd] built {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:     # python -m consistentid_amd.build --variant NAME DEF[=V] ...
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    else:
        build(force="--force" in sys.argv)
