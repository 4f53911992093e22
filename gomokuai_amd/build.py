"""Builds libgomoku_hip.so (HIP kernels + C-ABI) and the CorePyExt pybind11 module for gfx950, in-tree."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgomoku_hip.so")
# The profiling flavour (-DGMK_PROFILE): the same sources with the phase masks / in-kernel timers / launch-shape overrides that
# tools/*.sh drive through the environment compiled IN; the production library has none of them.  Loaded only when asked for
# (GMK_HIP_LIB=prof, see lib.py); never by the tests, bench.py or __graft_entry__.
PROF_LIB = os.path.join(HERE, "libgomoku_hip_prof.so")

LIB_SOURCES = ["capi.hip", "eval_kernel.hip", "evalstate_kernel.hip", "trad_kernel.hip", "rave_kernel.hip", "az_kernel.hip", "pvnet_kernel.hip", "mcts_kernel.hip", "records_kernel.hip", "pattern_tables.cpp", "synth.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: MCTS numerics (f64 PUCB from f32 operands, f32 running mean) must match the CPU
# restatement bit for bit; hipcc fuses multiply-add by default.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = list(sources) + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "gomoku_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False, profile=False):
    lib = PROF_LIB if profile else LIB
    objdir = os.path.join(CSRC, "prof") if profile else CSRC
    os.makedirs(objdir, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not (force or _stale(lib, srcs)):
        return lib
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.splitext(os.path.basename(s))[0] + ".o")
        if force or _stale(o, [s]):
            cmd = [HIPCC] + FLAGS + (["-DGMK_PROFILE"] if profile else []) + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


def build_pyext(force=False, verbose=False):
    src = os.path.join(CSRC, "core_pyext.cpp")
    if not os.path.exists(src):
        return None
    import pybind11
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(HERE, "CorePyExt" + suffix)
    if not (force or _stale(out, [src, LIB])):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off",
           "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
           src, "-o", out, "-L" + HERE, "-lgomoku_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    lib = build_lib(force, verbose)
    ext = build_pyext(force, verbose)
    return lib, ext


if __name__ == "__main__":
    if "--profile" in sys.argv:
        print(build_lib(force="--force" in sys.argv, verbose=True, profile=True))
    else:
        print(build_all(force="--force" in sys.argv, verbose=True))
