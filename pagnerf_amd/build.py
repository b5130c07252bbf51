"""Build libpagnerf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m pagnerf_amd.build [--force]

Objects land in pagnerf_amd/lib/obj/, the library in pagnerf_amd/lib/libpagnerf_hip.so (in-tree so
that it travels to the GPU box; both are git-ignored).  A source/flag hash makes rebuilds a no-op.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpagnerf_hip.so")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-value"] + \
    os.environ.get("PAG_EXTRA_FLAGS", "").split()          # kernel experiments only (e.g. -DPAG_DBG_...)
# encode / render reproduce the oracle's fp32 op order: no FMA contraction there
SOURCES = {
    "api.cpp": ["-x", "hip"],
    "encode.hip": ["-ffp-contract=off"],
    "render.hip": ["-ffp-contract=off"],
    # MFMA results in VGPRs wherever they fit: the fused backward kernels keep 224 accumulator registers in the AGPR half, and with the
    # default (AGPR-form MFMAs for the whole function) every per-tile result paid a v_accvgpr_read per register - 160 of 520 VALU per tile
    "mlp.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "assign.hip": ["-ffp-contract=off"],
    "loss.hip": ["-ffp-contract=off"],
    "optim.hip": ["-ffp-contract=off"],      # Adam: torch's op order, no contraction beyond the explicit fmaf
    "pose.hip": ["-ffp-contract=off"],       # the camera transform in the tensor-op form's op order
    "regularizer.hip": ["-ffp-contract=off"],
    "sparse.hip": [],                         # the touched-rows exchange's mask / plan / pack / unpack passes: integer and copy work only
}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


_COMPILER_ID = None


def _compiler_id():
    """The HIP / clang version lines of `hipcc --version`: part of the digest, so that a library built by another toolchain is not
    taken for up to date (the paths hipcc also prints are left out - they differ between boxes of one image)."""
    global _COMPILER_ID
    if _COMPILER_ID is None:
        try:
            out = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, timeout=120).stdout
        except Exception as e:              # no compiler: the digest then cannot match a stamp written where one was present
            out = "unavailable: %r" % (e,)
        _COMPILER_ID = "\n".join(l.strip() for l in out.splitlines() if l.startswith(("HIP version", "AMD clang version"))) or out.strip()
    return _COMPILER_ID


def _digest():
    h = hashlib.sha256()
    h.update(repr((COMMON, sorted(SOURCES.items()))).encode())
    h.update(_compiler_id().encode())
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in sorted(os.listdir(root)):
            if not os.path.isfile(os.path.join(root, f)):
                continue
            with open(os.path.join(root, f), "rb") as fh:
                h.update(f.encode())
                h.update(fh.read())
    return h.hexdigest()


def _compile(src, flags, obj):
    cmd = [_hipcc()] + COMMON + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-4000:]))
    return src


def build(force=False, verbose=True):
    os.makedirs(os.path.join(LIBDIR, "obj"), exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        if verbose:
            print("[pagnerf_amd.build] up to date:", LIB, file=sys.stderr)
        return LIB
    objs = {s: os.path.join(LIBDIR, "obj", os.path.splitext(s)[0] + ".o") for s in SOURCES}
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        futs = [ex.submit(_compile, s, f, objs[s]) for s, f in SOURCES.items()]
        for f in concurrent.futures.as_completed(futs):
            if verbose:
                print("[pagnerf_amd.build] compiled", f.result(), file=sys.stderr)
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + list(objs.values())
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print("[pagnerf_amd.build] linked", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
