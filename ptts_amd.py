"""Importable alias for the package directory ``personalized_text-to-speech_amd`` (its name has a
hyphen, so ``import`` cannot spell it): ``import ptts_amd`` yields that package object."""
import importlib
import sys

_pkg = importlib.import_module("personalized_text-to-speech_amd")
sys.modules[__name__] = _pkg
