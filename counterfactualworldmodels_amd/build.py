"""Build libcwm_hip.so (hipcc, gfx950) in-tree.  `python -m counterfactualworldmodels_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcwm_hip.so")
DEV_LIB_PATH = os.path.join(LIB_DIR, "libcwm_hip_dev.so")  # the same objects + csrc/dev.hip (switches, micro-benchmarks: include/cwm_hip_dev.h)
DEV_SOURCES = ["dev.hip"]
SOURCES = ["gemm.hip", "attention.hip", "attention_pipe.hip", "elementwise.hip", "conj_kernels.hip", "conj_attention.hip", "flowstats.hip", "engine.hip", "model.hip",
           "conj_model.hip", "comm.hip"]
HEADERS = ["exports.map", "common.h", "kernels.h", "gemm_device.h", "attention_device.h", "attention_tail.h", "engine.h", os.path.join("..", "..", "include", "cwm_hip.h"),
           os.path.join("..", "..", "include", "cwm_hip_dev.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def hipcc_version() -> str:
    """`HIP version ...; AMD clang version ...` of the compiler a build would use (compiled into the library: cwm_compiler_version())."""
    out = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if l.startswith("HIP version") or "clang version" in l]
    return "; ".join(keep) or "unknown"


LINT_RECORD = os.path.join(CSRC, "LINT_PASSED.json")


def lint_record() -> dict:
    """What tools/asm_lds_lint.py --record last wrote: the compiler version the ISA lint of the hand-counted waits passed on .  (The lint itself runs on the current sources in the CPU test suite; the record pins the COMPILER.)  The kernels' counted `s_waitcnt`s are only as good as the ISA the compiler emits around them."""
    import json

    try:
        with open(LINT_RECORD) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def check_lint_record(verbose: bool = True) -> bool:
    """True when this compiler is the one the ISA lint last passed on; warns otherwise (a toolchain bump can move instructions across the hand-counted
    waits without any test noticing: re-run `python tools/asm_lds_lint.py --record`)."""
    rec = lint_record()
    ok = rec.get("hipcc") == hipcc_version()
    if not ok and verbose:
        print("WARNING: hipcc is `%s` but the ISA lint of the hand-counted waits last passed on `%s`: run `python tools/asm_lds_lint.py --record` before trusting "
              "this build" % (hipcc_version(), rec.get("hipcc", "<never recorded>")), file=sys.stderr)
    return ok


def source_hash() -> str:
    """sha1 over every source and header of the library (what `cwm_source_hash()` of a current build returns)."""
    import hashlib

    h = hashlib.sha1()
    for f in sorted(SOURCES + DEV_SOURCES + HEADERS):
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stamp_hash() -> str:
    try:
        with open(os.path.join(PKG_DIR, "build", "prod", "source_hash.txt")) as fh:
            return fh.read().strip()
    except OSError:
        return ""


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH) or not os.path.exists(DEV_LIB_PATH):
        return True
    t = min(os.path.getmtime(LIB_PATH), os.path.getmtime(DEV_LIB_PATH))
    deps = [os.path.join(CSRC, s) for s in SOURCES + DEV_SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps) or _stamp_hash() != source_hash()


def _compile_one(args):
    cmd, obj = args
    res = subprocess.run(cmd, capture_output=True, text=True)
    return obj, res.returncode, res.stdout + res.stderr


def build_library(force: bool = False, verbose: bool = False, out_path: str = None) -> str:
    """Compile every source to an object (in parallel, only those older than a dependency) and link.  `out_path` (or the
    CWM_HIP_LIB_OUT environment variable) builds a side library -- e.g. a -DCWM_ATTN_PROF profiling build -- without replacing
    the production one."""
    out_path = out_path or os.environ.get("CWM_HIP_LIB_OUT") or LIB_PATH
    extra = os.environ.get("CWM_HIPCC_EXTRA", "").split()  # e.g. -DCWM_ATTN_PROF (phase timers in attention_pipe.hip)
    if extra and out_path == LIB_PATH:
        raise RuntimeError("CWM_HIPCC_EXTRA builds must not overwrite the production library: set CWM_HIP_LIB_OUT (and load it with CWM_HIP_LIB)")
    if not force and out_path == LIB_PATH and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    import hashlib

    # (hashlib, not hash(): the latter is salted per process, so side builds never found their objects again)
    tag = "prod" if out_path == LIB_PATH else "side_" + hashlib.sha1(" ".join(extra + [os.path.abspath(out_path)]).encode()).hexdigest()[:8]
    obj_dir = os.path.join(PKG_DIR, "build", tag)
    os.makedirs(obj_dir, exist_ok=True)
    # One builder at a time per object directory: under torchrun every rank that finds the library stale lands here at once and
    # they would write the same objects and the same link output.  The lock is held over compile + link + replace; a rank that
    # waited re-checks and usually finds the work done.
    import fcntl

    with open(os.path.join(obj_dir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and out_path == LIB_PATH and not needs_build():
                return LIB_PATH
            return _build_locked(out_path, extra, obj_dir, force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(out_path, extra, obj_dir, force, verbose):
    # -fvisibility=hidden: the library exports the C ABI of include/cwm_hip.h (CWM_API) and nothing else
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"] + extra
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    jobs, objs = [], []
    stamp = os.path.join(obj_dir, "source_hash.txt")
    shash = source_hash()
    try:
        with open(stamp) as fh:
            stale_hash = fh.read().strip() != shash
    except OSError:
        stale_hash = True
    dev_objs = [os.path.join(obj_dir, s + ".o") for s in DEV_SOURCES]
    for s in SOURCES + DEV_SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(obj_dir, s + ".o")
        if s in SOURCES:
            objs.append(obj)
        if s == "engine.hip":  # carries the hash of ALL sources (cwm_source_hash): recompiled whenever anything changed
            # (`engine.hip.o.hash` = the hash THIS object was compiled with: after a build that failed on another source the object already carries the new hash while
            # the stamp still holds the old one -- reverting the edit then made "stamp == sources" true and linked an object with the wrong hash)
            try:
                with open(obj + ".hash") as fh:
                    obj_hash = fh.read().strip()
            except OSError:
                obj_hash = ""
            if force or stale_hash or obj_hash != shash or not os.path.exists(obj):
                jobs.append((base + ['-DCWM_SRC_HASH="%s"' % shash, '-DCWM_HIPCC_VERSION="%s"' % hipcc_version().replace('"', "'"), "-c", src, "-o", obj], obj))
                try:
                    os.remove(obj + ".hash")
                except OSError:
                    pass
            continue
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(src)):
            jobs.append((base + ["-c", src, "-o", obj], obj))
    if verbose:
        print("compiling %d of %d sources" % (len(jobs), len(SOURCES)), file=sys.stderr)
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as ex:
            for obj, rc, log in ex.map(_compile_one, jobs):
                if rc != 0:
                    raise RuntimeError("hipcc failed on %s:\n%s" % (obj, log))
                if obj.endswith("engine.hip.o"):
                    with open(obj + ".hash", "w") as fh:
                        fh.write(shash)
                if verbose and log.strip():
                    print(log, file=sys.stderr)
    tmp = "%s.tmp%d" % (out_path, os.getpid())
    link = ["-shared", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]
    res = subprocess.run(base + link + objs + ["-ldl", "-o", tmp], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, out_path)
    # the development library: the production objects + dev.hip, beside the production one (side builds: <name>_dev.so)
    dev_out = DEV_LIB_PATH if out_path == LIB_PATH else out_path[:-3] + "_dev.so"
    tmp = "%s.tmp%d" % (dev_out, os.getpid())
    res = subprocess.run(base + link + objs + dev_objs + ["-ldl", "-o", tmp], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed (development library):\n" + res.stdout + res.stderr)
    os.replace(tmp, dev_out)
    with open(stamp, "w") as fh:
        fh.write(shash)
    if force and out_path == LIB_PATH:
        clean_stale()
    check_lint_record()
    return out_path


def clean_stale() -> list:
    """`--force` on the production library also clears what earlier rounds left under build/: objects of sources that no longer exist and the
    `side_*` directories of one-off side builds (CWM_HIPCC_EXTRA profiling builds) -- they travelled with every push to the GPU box.  Returns what it removed."""
    root = os.path.join(PKG_DIR, "build")
    gone = []
    keep = {s + ".o" for s in SOURCES + DEV_SOURCES} | {"source_hash.txt", ".lock", "engine.hip.o.hash"}
    prod = os.path.join(root, "prod")
    if os.path.isdir(prod):
        for f in os.listdir(prod):
            if f not in keep:
                os.remove(os.path.join(prod, f))
                gone.append(os.path.join("prod", f))
    if os.path.isdir(root):
        for d in os.listdir(root):
            if d.startswith("side_"):
                shutil.rmtree(os.path.join(root, d), ignore_errors=True)
                gone.append(d)
    return gone


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
