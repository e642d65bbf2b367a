"""Import shim: ``import isubgvqa_amd`` loads the package that lives in the (non-identifier) directory
``intrinsic-subgraph-generation-for-vqa_amd/``."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "intrinsic-subgraph-generation-for-vqa_amd")
_spec = importlib.util.spec_from_file_location("isubgvqa_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["isubgvqa_amd"] = _mod
_spec.loader.exec_module(_mod)
