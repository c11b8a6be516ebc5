"""Build libddp_hip.so in-tree for gfx950:  python -m diffdock_pocket_amd.build
(hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = ["ddp_conv.hip", "ddp_misc.hip", "ddp_gemm.hip", "ddp_pose.hip", "ddp_graph.hip", "ddp_views.hip", "ddp_capi.hip"]
OUT = os.path.join(HERE, "libddp_hip.so")


def _deps():
    return [os.path.join(HERE, "csrc", s) for s in SOURCES] + [os.path.join(HERE, "csrc", "ddp_internal.h"),
                                                               os.path.join(ROOT, "include", "ddp_hip.h")]


def source_hash():
    """16 hex digits of the SHA-256 over every source the library is compiled from, in a fixed order.  Compiled into the
    library (-DDDP_SRC_SHA16, exported as ddp_source_hash()) and checked by _lib.load(): the binary that runs is provably
    the one built from the sources in the tree."""
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_hash(path=None):
    """ddp_source_hash() of an existing library (None if it is missing or predates the export)."""
    import ctypes
    path = path or OUT
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        lib.ddp_source_hash.restype = ctypes.c_char_p
        return lib.ddp_source_hash().decode()
    except (OSError, AttributeError):
        return None


def needs_build():
    """The library is rebuilt unless it carries the hash of the present sources (mtimes are not trusted: the .so travels to
    the GPU box with the snapshot)."""
    return built_hash() != source_hash()


def build(force=False, verbose=True, stamps=False, ablate=0, defs=(), tag=""):
    if defs:   # tuning variants (tools/): extra -D flags, library name suffixed with `tag`
        out = os.path.join(HERE, f"libddp_hip_{tag}.so")
        cmd = [os.environ.get("HIPCC", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-o", out]
        cmd += [f"-D{d}" for d in defs] + [os.path.join(HERE, "csrc", s) for s in SOURCES]
        subprocess.check_call(cmd)
        return out
    if ablate or stamps:
        # diagnostic variants, never loaded by the product: -DDDP_STAMPS = in-kernel phase stamps (tools/stamp_conv.py),
        # -DDDP_ABLATE=n = timing-only ablations whose results are wrong by construction (tools/ablate_conv.py)
        out = os.path.join(HERE, "libddp_hip" + ("_stamps" if stamps else "") + (f"_ablate{ablate}" if ablate else "") + ".so")
        cmd = [os.environ.get("HIPCC", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-o", out]
        cmd += (["-DDDP_STAMPS"] if stamps else []) + ([f"-DDDP_ABLATE={ablate}"] if ablate else [])
        cmd += [os.path.join(HERE, "csrc", s) for s in SOURCES]
        subprocess.check_call(cmd)
        return out
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", f'-DDDP_SRC_SHA16="{source_hash()}"',
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-o", OUT]
    cmd += [os.path.join(HERE, "csrc", s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, stamps="--stamps" in sys.argv)
