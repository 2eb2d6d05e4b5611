"""Builds libdsc_hip.so (hipcc, gfx950) in-tree.  Used by __graft_entry__.build() and usable standalone:
    python -m diffusionspatialcontrol_amd.build
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdsc_hip.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17"]
# the hipBLASLt fallback (csrc/linear_lt.hip) is a BUILD option, off by default: the default .so neither contains nor links the
# library.  DSC_WITH_HIPBLASLT=1 in the environment of the build compiles it in (A/B runs against the package's own GEMMs).
WITH_HIPBLASLT = os.environ.get("DSC_WITH_HIPBLASLT", "0") not in ("", "0")
if WITH_HIPBLASLT:
    FLAGS = FLAGS + ["-DDSC_WITH_HIPBLASLT=1"]


STAMP = os.path.join(HERE, "_obj", "sources.sha256")


def source_hash():
    """sha256 over the names and bytes of every file the library is built from (csrc/*, include/dsc_hip.h) and the compiler
    flags: a stale .so that merely has a newer mtime than the sources (a checkout, a copied tree) is rebuilt"""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = sorted(f for f in glob.glob(os.path.join(CSRC, "*")) if os.path.isfile(f)) + [os.path.join(ROOT, "include", "dsc_hip.h")]
    for d in deps:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    objs = []
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(HERE, "_obj", os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        cmd = [hipcc] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + (["-lhipblaslt"] if WITH_HIPBLASLT else [])
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
