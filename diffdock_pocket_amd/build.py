"""Build libddp_hip.so in-tree for gfx950:  python -m diffdock_pocket_amd.build
(hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = ["ddp_conv.hip", "ddp_misc.hip", "ddp_gemm.hip", "ddp_pose.hip", "ddp_graph.hip", "ddp_views.hip", "ddp_capi.hip"]
OUT = os.path.join(HERE, "libddp_hip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, "csrc", s) for s in SOURCES] + [os.path.join(HERE, "csrc", "ddp_internal.h"),
                                                               os.path.join(ROOT, "include", "ddp_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


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
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-o", OUT]
    cmd += [os.path.join(HERE, "csrc", s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, stamps="--stamps" in sys.argv)
