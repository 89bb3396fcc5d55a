"""Fall-through to the reference checkout for the parts of ``reconstruction`` this build does not replace.

With ``CINE_REFERENCE_ROOT`` set, ``extend_path`` appends the reference's directory of the same sub-package to a
package's ``__path__`` (modules this build ships win, everything else -- ``data.mri_data``, ``data.subsample``,
``data.volume_sampler``, ``pl_modules`` ... -- resolves to the reference), and ``load_shadowed`` loads the reference
module that a build module of the same name shadows (``data/transforms.py``) under a private name so its remaining
attributes can be forwarded.
"""
import importlib.util
import os
import sys


def reference_dir(*parts: str):
    root = os.environ.get("CINE_REFERENCE_ROOT")
    if not root:
        return None
    d = os.path.join(root, "reconstruction", *parts)
    return d if os.path.exists(d) else None


def extend_path(path_list, *parts: str) -> None:
    d = reference_dir(*parts)
    if d and os.path.isdir(d) and d not in path_list:
        path_list.append(d)


def load_shadowed(package: str, module: str):
    """Import ``$CINE_REFERENCE_ROOT/reconstruction/<package tail>/<module>.py`` as ``<package>._reference_<module>``."""
    name = f"{package}._reference_{module}"
    if name in sys.modules:
        return sys.modules[name]
    tail = package.split(".")[1:]
    path = reference_dir(*tail, module + ".py")
    if path is None:
        raise ImportError(f"{package}.{module}: this attribute lives in the reference checkout; set CINE_REFERENCE_ROOT "
                          "to f78bono/deep-cine-cardiac-mri (see INTEGRATION.md)")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(name, None)
        raise
    return mod
