"""``import mmee_amd`` — importable alias of the package directory ``multi-modal-early-exit_amd`` (a hyphenated
directory name cannot appear in an ``import`` statement)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("multi-modal-early-exit_amd")
