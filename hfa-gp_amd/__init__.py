"""hfa_gp_amd — MI355X-native hot path of HFA-GP (EG3D tri-plane generator + latent-basis layer).

Importable as ``hfa_gp_amd`` (the directory is ``hfa-gp_amd/``; ``hfa_gp_amd/__init__.py``
at the repo root points the import system here).
"""
from .config import GeneratorConfig, PRESETS, ffhq512_128, small128, tiny14, tiny64  # noqa: F401

__all__ = ["GeneratorConfig", "PRESETS", "ffhq512_128", "tiny64", "small128"]
