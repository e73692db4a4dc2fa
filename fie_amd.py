"""Import alias: the product package lives in the directory
``fast-image-editing-with-generative-models_amd/`` (a hyphenated name Python cannot
import directly).  ``import fie_amd`` loads that directory as the package ``fie_amd``.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "fast-image-editing-with-generative-models_amd")
_spec = importlib.util.spec_from_file_location(
    "fie_amd", os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fie_amd"] = _mod
_spec.loader.exec_module(_mod)
