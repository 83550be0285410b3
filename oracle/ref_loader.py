"""
Loaders for the REAL reference.  TEST INFRASTRUCTURE ONLY (see oracle/ed_oracle.c header).

* ``load_ref_ext()``  -- the reference's C extension ``_deform_grid`` as compiled by
  ``make -C oracle ref`` into oracle/_ref/ (binary only; travels to the GPU box).  Exposes
  ``deform_grid``, ``deform_grid_grad``, ``spline_filter1d_grad`` (_deform_grid.c:306-311).
  Returns None when the binary is absent.
* ``load_reference()`` -- the full reference package (its Python layer + that extension).  Only
  possible where /root/reference exists, i.e. in the build container: the reference's Python
  sources never leave it.  A throw-away package directory of symlinks is assembled under /tmp.
  Returns None when /root/reference or the binary is absent.
"""
import glob
import importlib
import importlib.util
import os
import sys
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("ED_REFERENCE_ROOT", "/root/reference")


def _ext_path():
    hits = sorted(glob.glob(os.path.join(_HERE, "_ref", "_deform_grid*.so")))
    return hits[0] if hits else None


def load_ref_ext():
    path = _ext_path()
    if path is None:
        return None
    name = "_deform_grid"
    if name in sys.modules and getattr(sys.modules[name], "__file__", None) == path:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_ref_pkg = None


def load_reference():
    global _ref_pkg
    if _ref_pkg is not None:
        return _ref_pkg
    src = os.path.join(REF_ROOT, "elasticdeform")
    ext = _ext_path()
    if ext is None or not os.path.isdir(src):
        return None
    root = tempfile.mkdtemp(prefix="ed_ref_pkg_")
    # (imported under a private name: `elasticdeform` itself is this repo's drop-in alias package)
    pkg = os.path.join(root, "ed_reference_pkg")
    os.mkdir(pkg)
    for f in ("__init__.py", "deform_grid.py", "torch.py"):
        os.symlink(os.path.join(src, f), os.path.join(pkg, f))
    os.symlink(ext, os.path.join(pkg, os.path.basename(ext)))
    sys.path.insert(0, root)
    try:
        _ref_pkg = importlib.import_module("ed_reference_pkg")
    finally:
        sys.path.remove(root)
    return _ref_pkg
