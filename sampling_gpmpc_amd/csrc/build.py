#!/usr/bin/env python3
"""Build libgpmpc_hip.so for gfx950 in-tree with hipcc (cross-compiles without a GPU).

    python sampling_gpmpc_amd/csrc/build.py [--force] [--verbose]
"""
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["capi.hip", "rollout.hip", "rollout_fast.hip", "rollout_tiles.hip", "rollout_one.hip", "rollout_indep.hip", "joint.hip",
           "joint_mfma.hip", "joint_chol.hip", "assemble.hip", "base_samples.hip"]
# everything a source may include: the generated statement files (.inc) count like headers - editing a generator's OUTPUT
# rebuilds the kernels that include it; tests/test_generated_sources.py checks that the committed .inc files are what the
# generators (tools/gen_rollout_one.py, tools/gen_mfma_chains.py) produce
HEADERS = ["gpmpc_device.hpp", "gpmpc_host.hpp", "rollout_args.hpp", "joint_args.hpp", "joint_eigh.hpp",
           "rollout_one_gen.inc", "rollout_tiles_mfma.inc", "joint_mfma_gen.inc",
           os.path.join(REPO, "include", "gpmpc_hip.h")]
GENERATED = {"rollout_one_gen.inc": os.path.join(REPO, "tools", "gen_rollout_one.py"),
             "rollout_tiles_mfma.inc": os.path.join(REPO, "tools", "gen_mfma_chains.py"),
             "joint_mfma_gen.inc": os.path.join(REPO, "tools", "gen_joint_mfma.py")}
OUT = os.path.join(os.path.dirname(HERE), "libgpmpc_hip.so")
OBJDIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fgpu-rdc" if False else "-fno-gpu-rdc",
         "-ffp-contract=on", "-I", os.path.join(REPO, "include"), "-I", HERE]
FLAGS += os.environ.get("GPMPC_EXTRA_DEFS", "").split()   # experiment knobs, e.g. "-DGPMPC_JOINT_WPE=4"
if os.environ.get("GPMPC_PHASE_TIMERS") == "1":      # debug build: per-phase s_memtime counters in the tuned kernel
    FLAGS.append("-DGPMPC_PHASE_TIMERS")


# Per-file code-generation flags.  rollout_fast.hip runs at one wave per SIMD: machine-LICM hoists ~100 registers of
# constants/addresses out of the time-step loop, which starves the scheduler of VGPRs and serialises the LDS loads.
# joint.hip sits at the 128-VGPR cliff: without machine-LICM its kernels spill 0-50 registers instead of 55-77 (2-9 % per
# joint draw, tools/bench_joint.py / bench.py extras, round 2).
EXTRA_FLAGS = {"rollout_fast.hip": os.environ.get("GPMPC_FAST_FLAGS", "-mllvm -disable-machine-licm").split(),
               "joint.hip": os.environ.get("GPMPC_JOINT_FLAGS", "-mllvm -disable-machine-licm").split(),
               # the generic rollout kernel: 119-152 VGPRs instead of 131-174 (car Ns=4096 H=40: 22.3 -> 13.2 ms; pendulum 1.14 ms
               # either way); the sample-per-lane kernels of rollout_indep.hip measure 0.093 / 0.543 ms with it off, 0.097 / 0.547 on
               "rollout.hip": os.environ.get("GPMPC_ROLLOUT_FLAGS", "-mllvm -disable-machine-licm").split(),
               "rollout_tiles.hip": os.environ.get("GPMPC_TILES_FLAGS", "").split(),
               # rollout_one.hip: without machine-LICM no SGPR is spilled (28 otherwise: v_readlane / v_writelane pairs in the step)
               "rollout_one.hip": os.environ.get("GPMPC_ONE_FLAGS", "-mllvm -disable-machine-licm").split(),
               "rollout_indep.hip": os.environ.get("GPMPC_INDEP_FLAGS", "").split(),
               "joint_mfma.hip": os.environ.get("GPMPC_JOINT_MFMA_FLAGS", "").split()}


STAMP = os.path.join(OBJDIR, "flags.stamp")


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def flags_digest():
    """Hash of everything besides the sources that decides what the objects contain: compiler, common flags (incl. the
    GPMPC_EXTRA_DEFS / GPMPC_PHASE_TIMERS experiment knobs) and the per-file flags.  An ablation or instrumentation
    build (tools/*.sh) therefore never survives as the "up to date" library of a later default build."""
    blob = json.dumps({"hipcc": HIPCC, "flags": FLAGS, "extra": EXTRA_FLAGS}, sort_keys=True)
    return hashlib.sha256(blob.encode()).hexdigest()


def is_default_build():
    """True when no experiment knob is set in the environment (what tests, bench.py and build() expect to load)."""
    return not (os.environ.get("GPMPC_EXTRA_DEFS") or os.environ.get("GPMPC_PHASE_TIMERS") == "1"
                or os.environ.get("GPMPC_FAST_FLAGS") is not None)


def clean_objdir():
    """Only objects and the flags stamp belong in csrc/build/: -save-temps leftovers (.hipi, .bc, .s, .out) of experiment
    builds would travel to the GPU box with every snapshot."""
    if not os.path.isdir(OBJDIR):
        return
    for name in os.listdir(OBJDIR):
        p = os.path.join(OBJDIR, name)
        if os.path.isfile(p) and not (name.endswith(".o") and "-hip-amdgcn" not in name) and name != "flags.stamp":
            os.remove(p)


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    clean_objdir()
    digest = flags_digest()
    try:
        with open(STAMP) as f:
            stale_flags = f.read().strip() != digest
    except OSError:
        stale_flags = True
    force = force or stale_flags
    hdr_t = max(_mtime(h if os.path.isabs(h) else os.path.join(HERE, h)) for h in HEADERS)
    hdr_t = max(hdr_t, _mtime(os.path.abspath(__file__)))
    jobs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _mtime(o) < max(_mtime(s), hdr_t):
            jobs.append((s, o))
    def cc(job):
        s, o = job
        cmd = [HIPCC, "-x", "hip", "-c", s, "-o", o] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), [])
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    with open(STAMP, "w") as f:
        f.write(digest + "\n")
    objs = [os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(OUT):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    out = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(out)
