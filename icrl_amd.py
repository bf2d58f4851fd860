"""Importable alias for the hyphen-named package directory:
``import icrl_amd; icrl_amd.trainers.a2c_training(...)``."""
import importlib as _importlib

_PKG = "image-captioning-through-rl_amd"
synthetic = _importlib.import_module(_PKG + ".synthetic")
hipabi = _importlib.import_module(_PKG + ".hipabi")
models = _importlib.import_module(_PKG + ".models")
engine = _importlib.import_module(_PKG + ".engine")
optim = _importlib.import_module(_PKG + ".optim")
dist = _importlib.import_module(_PKG + ".dist")
utilities = _importlib.import_module(_PKG + ".utilities")
trainers = _importlib.import_module(_PKG + ".trainers")
