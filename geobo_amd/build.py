"""Build libgeobo_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m geobo_amd.build            # or: from geobo_amd.build import build; build()

The shared object is written IN-TREE (geobo_amd/lib/libgeobo_hip.so) so that it travels with the
repository snapshot to the GPU box; it is git-ignored.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgeobo_hip.so")
SOURCES = ["gemm_f64.hip", "gemm_fold.hip", "potrf.hip", "assembly.hip", "toeplitz.hip", "spectral_y.hip", "xz2d.hip", "xz2d_fold.hip", "reduce.hip"]
HEADERS = [os.path.join(CSRC, "covfun.h"), os.path.join(ROOT, "include", "geobo_hip.h")]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built on this machine")


def _deps(src):
    return [os.path.join(CSRC, src)] + HEADERS


def _flag_key(extra_flags=()):
    """Objects are cached per flag set: a build with extra flags (debug, -D switches) never feeds a later plain build."""
    import hashlib
    return "default" if not extra_flags else hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:12]


def _obj(src, extra_flags=()):
    return os.path.join(LIBDIR, "obj", _flag_key(extra_flags), os.path.splitext(src)[0] + ".o")


STAMP = LIB + ".flags"      # flag key of the objects the library was linked from


def up_to_date():
    """The in-tree library exists, was linked from default-flag objects and is newer than every source."""
    if not os.path.exists(LIB):
        return False
    if os.path.exists(STAMP) and open(STAMP).read().strip() != "default":
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(d) <= t for s in SOURCES for d in _deps(s))


def build(force=False, verbose=False, extra_flags=()):
    """Compile every HIP source of the package for gfx950 (one object per source, stale ones only, in parallel) and link them
    into one shared library."""
    if not force and not extra_flags and up_to_date():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(os.path.join(LIBDIR, "obj", _flag_key(extra_flags)), exist_ok=True)
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *extra_flags]

    def compile_one(src):
        obj = _obj(src, extra_flags)
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in _deps(src)):
            return None
        cmd = base + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        return subprocess.run(cmd, capture_output=True, text=True)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    for r in results:
        if r is not None and r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc failed building libgeobo_hip.so")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *[_obj(s, extra_flags) for s in SOURCES], "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking libgeobo_hip.so")
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        f.write(_flag_key(extra_flags) + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
