"""Build recipe for libequss_hip.so: plain ``hipcc --offload-arch=gfx950`` on csrc/*.hip, in-tree.

hipcc cross-compiles without a GPU, so the library is built in the CPU container and travels to
the GPU box with the source tree.  No torch.utils.cpp_extension (it hipifies CUDA sources) and no
CMake: the library has no dependency besides the HIP runtime.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_NAME = "libequss_hip.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
OBJ_DIR = os.path.join(PKG_DIR, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
         "-ffp-contract=on"]
# Per-file additions.  -fno-slp-vectorize for the attention kernel: its softmax sits between MFMAs,
# where the packed fp32 ops (v_pk_add_f32 / v_pk_mul_f32) the SLP vectoriser forms issue slower than
# the scalar pairs they replace (measured: 315 -> 288 us per launch).  The GEMM epilogues are faster
# with it left on.
FILE_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(PKG_DIR), "include", "equss_hip.h"))
    return hdrs


def source_digest():
    """sha256 (first 16 hex digits) over the kernel sources and headers, in name order: identifies the
    code a measurement (profiles/pmc_traffic.json) was taken on."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted([os.path.join(CSRC, f) for f in _sources()] + _headers()):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, jobs=None):
    """Compile every csrc/*.hip for gfx950 and link libequss_hip.so next to this file."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = _headers()
    todo, objs = [], []
    for src in _sources():
        obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + hdrs):
            todo.append((os.path.join(CSRC, src), obj))

    def compile_one(item):
        src, obj = item
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print("[equss build]", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{res.stdout}")
        return res.stdout

    jobs = jobs or min(8, max(1, len(todo)))
    if todo:
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            for out in ex.map(compile_one, todo):
                if verbose and out.strip():
                    print(out)
    if todo or force or _stale(LIB_PATH, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print("[equss build]", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}")
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB_PATH)
