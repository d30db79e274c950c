"""Compatibility namespace: the import paths the reference's drivers use
(`from ps_vae.model import VAEModel`, `from ps_vae.lightning import PseudoSpeakerVAE`, ...;
/root/reference/ps_vae/training.py:10, inference.py:1) resolve to the MI355X implementation."""
