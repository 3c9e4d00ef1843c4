"""Import alias: ``import video_gcp_amd`` loads the package that lives in ``video-gcp_amd/``.

The package directory carries the repository's required name (with a hyphen), which is not a
valid Python identifier; this one-file loader registers it under ``video_gcp_amd``.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "video-gcp_amd")
_spec = importlib.util.spec_from_file_location(
    "video_gcp_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["video_gcp_amd"] = _mod
_spec.loader.exec_module(_mod)
