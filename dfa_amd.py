"""Import alias: `import dfa_amd` loads the package kept in ./deep-fake-audio-classifier_amd/ (a directory name
Python cannot import directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "deep-fake-audio-classifier_amd")
_spec = importlib.util.spec_from_file_location("dfa_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dfa_amd"] = _mod
_spec.loader.exec_module(_mod)
