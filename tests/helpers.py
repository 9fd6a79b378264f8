"""Shared helpers for the test-suite (config loading, synthetic inputs of SURVEY.md section 8d live in the package)."""
import os

from sampling_gpmpc_amd.workloads import PARAMS_DIR as PARAMS, closed_loop_params, fs_params, load_params, synthetic_u_ff  # noqa: F401

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
