"""Imported first by every probe in tools/: points the package at the DIAGNOSTIC library (`make -C multi-modal-early-exit_amd/csrc diag`),
the only build that carries the stamped kernels, the timing variants (wrong results) and the A/B environment switches.  The release
library ignores all of them.  MMEE_LIB given by the caller wins."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
_lib = os.path.join(ROOT, "multi-modal-early-exit_amd", "libmmee_hip_diag.so")
if "MMEE_LIB" not in os.environ:
    if not os.path.exists(_lib):
        raise SystemExit(f"{_lib} is missing: make -C multi-modal-early-exit_amd/csrc diag")
    os.environ["MMEE_LIB"] = _lib
