"""The generated statement files under sampling_gpmpc_amd/csrc (*.inc) are what their generators produce, the build treats them
as dependencies, and the computed-jump tables of joint_mfma_gen.inc have their strides (CPU, hipcc cross-compiles)."""
import filecmp
import importlib.util
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "sampling_gpmpc_amd", "csrc")


def _build_module():
    spec = importlib.util.spec_from_file_location("gpmpc_build", os.path.join(CSRC, "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("inc", ["rollout_one_gen.inc", "rollout_tiles_mfma.inc", "joint_mfma_gen.inc"])
def test_committed_inc_is_the_generators_output(tmp_path, inc):
    b = _build_module()
    gen = b.GENERATED[inc]
    out = tmp_path / inc
    args = [sys.executable, gen, "--out", str(out)]
    r = subprocess.run(args, capture_output=True, text=True, cwd=REPO)
    if not out.exists() and r.returncode == 0 and r.stdout.strip():      # a generator that prints to stdout
        out.write_text(r.stdout)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(str(out), os.path.join(CSRC, inc), shallow=False), f"{inc} is not what {os.path.basename(gen)} produces: regenerate it"


def test_build_depends_on_generated_files_and_generators():
    b = _build_module()
    heads = [os.path.basename(h) for h in b.HEADERS]
    for inc, gen in b.GENERATED.items():
        assert inc in heads, f"{inc} is not a build dependency: editing it would not rebuild the kernels that include it"
        assert os.path.exists(gen)
        assert os.path.exists(os.path.join(CSRC, inc))


def test_build_directory_holds_objects_only(tmp_path):
    b = _build_module()
    junk = os.path.join(b.OBJDIR, "leftover-hip-amdgcn-amd-amdhsa-gfx950.hipi")
    os.makedirs(b.OBJDIR, exist_ok=True)
    open(junk, "w").write("x")
    b.clean_objdir()
    assert not os.path.exists(junk)


def test_joint_mfma_jump_tables_have_their_strides():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_joint_mfma_tables.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_hipcc_never_touches_the_pinned_agprs_of_rollout_one():
    """rollout_one_kernel's factor panels live in AGPRs that hipcc only knows through physical-register asm constraints, with some
    writes hidden under branches: the ISA hipcc emits with the build's flags must not name an AGPR outside the asm
    statements, and must not spill inside the kernel (tools/check_one_agpr.py)."""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_one_agpr.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
