"""Import helper: the package directory is ``tempenslearnagg-qcnet_amd`` (a hyphen is not a Python
identifier), importable as ``tempenslearnagg_qcnet_amd`` through the in-tree symlink or, if the
symlink did not survive a copy, through an explicit module spec."""
import importlib
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = 'tempenslearnagg_qcnet_amd'


def ensure_package():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    if PKG in sys.modules:
        return sys.modules[PKG]
    if os.path.isdir(os.path.join(ROOT, PKG)):
        return importlib.import_module(PKG)
    path = os.path.join(ROOT, 'tempenslearnagg-qcnet_amd')
    spec = importlib.util.spec_from_file_location(PKG, os.path.join(path, '__init__.py'),
                                                  submodule_search_locations=[path])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[PKG] = mod
    spec.loader.exec_module(mod)
    return mod
