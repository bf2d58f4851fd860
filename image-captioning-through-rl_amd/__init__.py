"""MI355X-native A2C caption-training hot path (gfx950 HIP kernels behind a C ABI).

The directory name carries a hyphen, so import it with
``importlib.import_module("image-captioning-through-rl_amd")`` or through the
``icrl_amd`` alias module at the repository root.  Submodules mirror the
reference's module names: ``models``, ``trainers``, ``utilities``.
"""
__all__ = ["synthetic"]
