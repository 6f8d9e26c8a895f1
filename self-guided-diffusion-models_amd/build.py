#!/usr/bin/env python3
"""Build libsgdm_hip.so (the C-ABI library of include/sgdm_hip.h) for gfx950, in-tree.

    python self-guided-diffusion-models_amd/build.py [--force]

hipcc cross-compiles without a GPU; objects land in csrc/build/, the shared library in
sgdm_amd/lib/libsgdm_hip.so (git-ignored, but it travels with the gpurun snapshot).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "sgdm_amd", "lib")
# tools: SGDM_BUILD_TAG=_x SGDM_EXTRA_FLAGS="-D..." builds libsgdm_hip_x.so from objects *_x.o next to the product library
TAG = os.environ.get("SGDM_BUILD_TAG", "")
LIB = os.path.join(LIBDIR, f"libsgdm_hip{TAG}.so")
# diagnostics (csrc/tools/*.hip, include/sgdm_hip_tools.h): a library of their own -- bench.py's device calibration, the
# contention tests and tools/ load it; the product path never does and the product library exports none of its symbols
TOOLS_SRC = os.path.join(CSRC, "tools")
TOOLS_LIB = os.path.join(LIBDIR, "libsgdm_hip_tools.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast"]
# per-file extras.  igemm: no SLP vectorisation -- packed f32 VALU (v_pk_fma_f32 ...) next to a saturated matrix pipe costs
# more issue time than the two scalar ops it replaces (measured +2..3 % on the conv launches with it off)
FILE_FLAGS = {"igemm.hip": ["-fno-slp-vectorize"], "pack.hip": ["-fno-slp-vectorize"]}      # (pack.hip: part of igemm.hip up to round 5)


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "sgdm_hip.h"))
    return hdrs


def _digest(paths, extra=()):
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def source_id():
    """identity of what the library is built FROM: every .hip / .h under csrc/, the public header and the compile
    flags.  Embedded in the library (sgd_build_id(), csrc/build_id.hip); __graft_entry__.build() compares the two, so a
    stale or foreign libsgdm_hip.so cannot pass for a build of this tree."""
    srcs = [os.path.join(CSRC, f) for f in _sources()]
    return _digest(srcs + _headers(), [FLAGS, sorted(FILE_FLAGS.items()), sorted(VARIANTS.items())])[:16]


# translation units compiled from ONE source with different defines (igemm.hip, the conv kernel: one unit per arithmetic mode,
# see the end of that file; its geometry / entry points are igemm_host.hip, the weight packing pack.hip) -- they build in parallel
# (*_nopk: the 1x1 / linear instances of the split modes once more with packed-f32 code generation off -- the LayerNorm-row
# prologue's launches run on these, csrc/igemm.hip: sgd_igemm_dispatch_*_nopk)
NOPK = ["-DSGDM_IGEMM_NOPK", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
VARIANTS = {"igemm.hip": [("_f32", ["-DSGDM_IGEMM_PREC=0"]), ("_f16x3", ["-DSGDM_IGEMM_PREC=1"]),
                          ("_bf16x3", ["-DSGDM_IGEMM_PREC=2"]), ("_f16x3_nopk", ["-DSGDM_IGEMM_PREC=1", *NOPK]),
                          ("_bf16x3_nopk", ["-DSGDM_IGEMM_PREC=2", *NOPK])]}


def _units():
    """(source, object suffix, extra flags) of every translation unit, longest compile first"""
    units = []
    for src in _sources():
        for suffix, extra in VARIANTS.get(src, [("", [])]):
            units.append((src, suffix, extra))
    return sorted(units, key=lambda u: (u[0] != "igemm.hip" or not u[1], u[0], u[1]))


def _compile(unit, force):
    src, suffix, extra = unit
    obj = os.path.join(OBJ, src[:-4] + suffix + TAG + ".o")
    sp = os.path.join(CSRC, src)
    cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(src, []), *extra, *(os.environ.get("SGDM_EXTRA_FLAGS", "").split() if TAG else []),
           "-c", sp, "-o", obj]
    if src == "build_id.hip":
        cmd.insert(1, f'-DSGDM_BUILD_ID="{source_id()}"')
    # an object is reused only when the CONTENT it was compiled from (source, headers, command line) is unchanged: the
    # digest sits next to it (mtimes say nothing after a checkout or a copy)
    want = _digest([sp] + _headers(), [cmd])
    stamp = obj + ".sha"
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return obj
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(want)
    return obj


def build_probe():
    """diagnostic variant of the conv kernel (ablation knobs + per-wave cycle stamps), tools/ only"""
    os.makedirs(OBJ, exist_ok=True)
    out = os.path.join(LIBDIR, "libsgdm_hip_probe.so")
    objs = []
    for src, suffix, extra in _units():
        obj = os.path.join(OBJ, src[:-4] + suffix + ".probe.o")
        r = subprocess.run([HIPCC, *FLAGS, *FILE_FLAGS.get(src, []), *extra, "-DSGDM_PROBE", *os.environ.get("SGDM_PROBE_FLAGS", "").split(), "-c",
                            os.path.join(CSRC, src), "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        objs.append(obj)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
    return out


def build_ablation(mask):
    """production kernels with one compile-time ablation mask (csrc/igemm.hip: SGDM_ABL), tools/ only"""
    os.makedirs(OBJ, exist_ok=True)
    out = os.path.join(LIBDIR, f"libsgdm_hip_abl{mask}.so")
    objs = []
    for unit in _units():
        src, suffix, extra = unit
        if src == "igemm.hip":
            obj = os.path.join(OBJ, f"igemm{suffix}.abl{mask}.o")
            r = subprocess.run([HIPCC, *FLAGS, *FILE_FLAGS.get(src, []), *extra, f"-DSGDM_ABL={mask}", *os.environ.get("SGDM_EXTRA_FLAGS", "").split(), "-c",
                                os.path.join(CSRC, src), "-o", obj],
                               capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(r.stderr)
        else:
            obj = _compile(unit, False)
        objs.append(obj)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
    return out


def build_tools(force=False):
    """libsgdm_hip_tools.so from csrc/tools/*.hip (content-digest reuse like the product objects)"""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    root = os.path.dirname(HERE)
    hdrs = [os.path.join(CSRC, "sgdm_common.h"), os.path.join(root, "include", "sgdm_hip_tools.h")]
    objs = []
    for f in sorted(x for x in os.listdir(TOOLS_SRC) if x.endswith(".hip")):
        sp, obj = os.path.join(TOOLS_SRC, f), os.path.join(OBJ, "tools_" + f[:-4] + ".o")
        cmd = [HIPCC, *FLAGS, "-c", sp, "-o", obj]
        want = _digest([sp] + hdrs, [cmd])
        stamp = obj + ".sha"
        if force or not os.path.exists(obj) or not os.path.exists(stamp) or open(stamp).read().strip() != want:
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for tools/{f}:\n{r.stdout}\n{r.stderr}")
            with open(stamp, "w") as fh:
                fh.write(want)
        objs.append(obj)
    want = _digest([o + ".sha" for o in objs])
    stamp = TOOLS_LIB + ".sha"
    if force or not os.path.exists(TOOLS_LIB) or not os.path.exists(stamp) or open(stamp).read().strip() != want:
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TOOLS_LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as fh:
            fh.write(want)
    return TOOLS_LIB


def build_lib(force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if not TAG:
        build_tools(force)
    with ThreadPoolExecutor(max_workers=int(os.environ.get("SGDM_BUILD_JOBS", "6"))) as ex:
        objs = list(ex.map(lambda u: _compile(u, force), _units()))
    want = _digest([o + ".sha" for o in objs])
    stamp = LIB + ".sha"
    if force or not os.path.exists(LIB) or not os.path.exists(stamp) or open(stamp).read().strip() != want:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(want)
    return LIB


if __name__ == "__main__":
    if "--abl" in sys.argv:
        print(build_ablation(int(sys.argv[sys.argv.index("--abl") + 1])))
    else:
        print(build_probe() if "--probe" in sys.argv else build_lib(force="--force" in sys.argv))
