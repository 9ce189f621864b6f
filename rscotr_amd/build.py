"""Build librscotr.so (HIP kernels + C ABI) in-tree for gfx950.

`python -m rscotr_amd.build` or `rscotr_amd.build.build_library()`. hipcc cross-compiles
without a GPU; the resulting .so is git-ignored but travels with the tree to the GPU box.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librscotr.so")
STAMP = os.path.join(HERE, "librscotr.so.stamp")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-fast-math",
         "-Wno-unused-result"]
# host-only sources (bit-exact fp64 restatements, e.g. the LSAP solver) must not be FMA-contracted
HOST_FLAGS = ["-ffp-contract=off"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith(".hip") or f.endswith(".cpp"))


def _digest():
    h = hashlib.sha256()
    for p in _sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
            [os.path.join(HERE, "..", "include", "rscotr.h")]:
        h.update(p.encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS + HOST_FLAGS).encode())
    return h.hexdigest()


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def build_library(force=False, verbose=True):
    """Compile every source under csrc/ into one shared object. Returns the .so path."""
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dig:
                return LIB
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build librscotr.so")
    objs = []
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    hdr = hashlib.sha256()
    for h in sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
            [os.path.join(HERE, "..", "include", "rscotr.h")]:
        with open(h, "rb") as fh:
            hdr.update(fh.read())
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        extra = HOST_FLAGS if src.endswith(".cpp") else []
        # per-object stamp: a source whose text, headers and flags are unchanged keeps its object
        with open(src, "rb") as fh:
            odig = hashlib.sha256(fh.read() + hdr.digest() + " ".join(FLAGS + extra).encode()).hexdigest()
        ostamp = obj + ".stamp"
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == odig:
            continue
        with open(ostamp, "w") as fh:
            fh.write(odig)
        cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, *extra, "-x", "hip", "-c", src, "-o", obj,
               "-I", CSRC, "-I", os.path.join(HERE, "..", "include")]
        if verbose:
            print("[rscotr build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            os.remove(os.path.join(objdir, os.path.basename(src) + ".o.stamp"))
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB]
    if verbose:
        print("[rscotr build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
