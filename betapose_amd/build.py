"""Build libbetapose_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m betapose_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbetapose_hip.so")
SOURCES = ["conv_igemm.hip", "conv_halo.hip", "conv_fused.hip", "conv_pl.hip", "conv_s1.hip", "conv_p3.hip", "aux_kernels.hip", "engine.cpp", "host_post.cpp", "frame_io.cpp", "jpeg_bmp.cpp", "c_api.cpp", "darknet_compat.cpp"]
HEADERS = ["bp_common.h", "engine.h", "frame_io.h", "conv_tail.inc", "conv_dev.h", os.path.join("..", "..", "include", "betapose_hip.h"),
           os.path.join("..", "..", "include", "yolo_v2_class_compat.h")]
# measured-and-superseded kernels (round-1/2 experiments) live in csrc/experimental/ and are compiled only into
# libbetapose_hip_exp.so (--experimental); the product library has no input under that directory
EXP = "experimental"
EXP_SOURCES = [os.path.join(EXP, s_) for s_ in ("conv_w64.hip", "conv_kg.hip", "conv_rd.hip")]
# ... and what only the experimental library includes (the persistent per-XCD launch and the unit that compiles it)
EXP_HEADERS = [os.path.join(EXP, s_) for s_ in ("mega.inc", "kernels_unity.hip")]
ARCH = "gfx950"
LAST_ACTION = None      # "compiled" | "reused": what the last build() of the product library did (__graft_entry__.build prints it)
# Host launch stubs every HIP object must export.  hipcc has been seen to drop a kernel's host stub SILENTLY (the object
# links, the launch then fails at run time) -- DESIGN.md §3.1d -- so the build counts them.
MIN_STUBS = {"conv_s1.hip": 1, "conv_p3.hip": 39, "conv_igemm.hip": 14, "conv_halo.hip": 13, "conv_fused.hip": 12, "conv_pl.hip": 12, "aux_kernels.hip": 17}
MIN_STUBS_EXP = {"conv_s1.hip": 1, "conv_p3.hip": 39, "experimental/kernels_unity.hip": 26, "conv_fused.hip": 12, "conv_pl.hip": 12, "experimental/conv_w64.hip": 8,
                 "experimental/conv_kg.hip": 3, "experimental/conv_rd.hip": 2, "aux_kernels.hip": 20}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libbetapose_hip.so)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]   # (product library: the experimental sources are not its inputs)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True, experimental: bool = False) -> str:
    """experimental: -DBP_EXPERIMENTAL into libbetapose_hip_exp.so (never loaded by the product: _lib.py BP_LIB)."""
    if experimental:
        return _build(True, verbose, os.path.join(HERE, "libbetapose_hip_exp.so"), os.path.join(HERE, "build_exp"),
                      ["-DBP_EXPERIMENTAL"])
    global LAST_ACTION
    if not force and not _stale():
        LAST_ACTION = "reused"
        return LIB
    LAST_ACTION = "compiled"
    return _build(force, verbose, LIB, os.path.join(HERE, "build"), [])


def _build(force: bool, verbose: bool, LIB: str, objdir: str, extra) -> str:
    cc = hipcc()
    objs = []
    os.makedirs(objdir, exist_ok=True)
    procs = []
    sources = SOURCES + (EXP_SOURCES if extra else [])
    if extra:   # the experimental library compiles the two conv files as ONE unit with the persistent per-XCD launch (mega.inc) behind them
        sources = [os.path.join(EXP, "kernels_unity.hip")] + [s_ for s_ in sources if s_ not in ("conv_igemm.hip", "conv_halo.hip")]
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS + (EXP_HEADERS if extra else []))
    flags_key = " ".join(list(extra) + os.environ.get("BP_CFLAGS", "").split())
    flags_file = os.path.join(objdir, ".flags")
    same_flags = os.path.exists(flags_file) and open(flags_file).read() == flags_key
    for src in sources:
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        # incremental: an object newer than its source and every header, built with the same flags, is kept (--force recompiles all)
        if not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            continue
        cmd = [cc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", "-I", CSRC,
               os.path.join(CSRC, src), "-o", obj] + list(extra) + os.environ.get("BP_CFLAGS", "").split()
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out)
    with open(flags_file, "w") as f:
        f.write(flags_key)
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    for src, want in (MIN_STUBS_EXP if extra else MIN_STUBS).items():
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        syms = subprocess.run([nm, obj], stdout=subprocess.PIPE, text=True).stdout
        have = syms.count("__device_stub__")
        if have < want:
            raise RuntimeError("%s: %d kernel host stubs in the object, expected >= %d (hipcc dropped a kernel)" % (src, have, want))
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experimental="--experimental" in sys.argv))
