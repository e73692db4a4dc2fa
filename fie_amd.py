"""Import alias: the product package lives in the directory
``fast-image-editing-with-generative-models_amd/`` (a hyphenated name Python cannot
import directly).  ``import fie_amd`` loads that directory as the package ``fie_amd``.
"""
import importlib.util
import os
import sys

# HIP deals streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues in order of first use.  One edit's graph already
# uses 2-3 streams (its forked branches), bench / run_batch keep 2 edits in flight, torch adds its capture stream: with 4
# queues two of them regularly share a queue and serialise (measured on MI355X: 2 edits in flight 11.1 instead of 14.0
# images/s; one serial edit 122 instead of 94 ms when the slot's stream collides with its own graph's branch).  16 queues
# give every stream this package creates its own queue.  The runtime reads the variable once when it initialises, so it is
# set here, before anything touches the GPU; an explicit setting in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "fast-image-editing-with-generative-models_amd")
_spec = importlib.util.spec_from_file_location(
    "fie_amd", os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fie_amd"] = _mod
_spec.loader.exec_module(_mod)
