"""Overlay support for the drop-in module tree (``dropin/``).

``dropin/`` holds ONLY the modules the HIP path replaces, at the reference's own import paths, as PEP 420 namespace
portions (no ``__init__.py`` anywhere -- the reference's ``dynamic`` / ``diffusion`` / ``dynamic_input`` packages have
none either).  With ``dropin/`` ahead of the reference checkout on ``sys.path`` Python merges both directories into one
package: a replaced module (``dynamic.diffusionmodules.openaimodel``) resolves here, every sibling the reference's own
code imports (``dynamic.attention_ldm``, ``dynamic_input.misc``, ``diffusion_utils.taokit.pl_utils`` ...) still resolves
in the checkout.

A replaced module defines only the names the HIP path implements.  Any OTHER name the reference imports from the same
module path (e.g. ``EncoderUNetModel`` in diffusion/classifier.py:13) is served lazily from the reference's own file of
that name, found further down the package's ``__path__`` -- see ``reference_fallback``.
"""
import importlib
import importlib.util
import os
import sys


def _reference_file(module_name, own_file):
    parent_name, _, leaf = module_name.rpartition(".")
    parent = importlib.import_module(parent_name)
    own_dir = os.path.dirname(os.path.abspath(own_file))
    for entry in list(getattr(parent, "__path__", [])):
        if os.path.abspath(entry) == own_dir:
            continue
        cand = os.path.join(entry, leaf + ".py")
        if os.path.isfile(cand):
            return cand
    return None


def reference_fallback(module_name, own_file):
    """-> a module-level ``__getattr__`` (PEP 562) that loads the shadowed reference module on first miss"""
    box = {}

    def __getattr__(name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if "mod" not in box:
            path = _reference_file(module_name, own_file)
            if path is None:
                box["mod"] = None
            else:
                alias = "_sgdm_reference." + module_name
                spec = importlib.util.spec_from_file_location(alias, path)
                mod = importlib.util.module_from_spec(spec)
                sys.modules[alias] = mod
                spec.loader.exec_module(mod)
                box["mod"] = mod
        ref = box["mod"]
        if ref is None or not hasattr(ref, name):
            raise AttributeError(f"module {module_name!r} has no attribute {name!r} (neither the MI355X drop-in nor a "
                                 "reference checkout behind it on sys.path defines it)")
        return getattr(ref, name)

    return __getattr__
