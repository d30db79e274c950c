"""Compatibility namespace: the import paths the reference's drivers use
(`from ps_vae.model import VAEModel`, `from ps_vae.lightning import PseudoSpeakerVAE`, ...;
/root/reference/ps_vae/training.py:8-10, inference.py:1) resolve to the MI355X implementation.

Only the hot path lives here.  Everything else of the reference's `ps_vae` package -- the loaders
(`ps_vae.data.cv`, `ps_vae.data.vctk`; training.py:8-9), `ps_vae.demographics`, the drivers -- keeps
resolving to the reference checkout: this package extends its `__path__` with every later `ps_vae`
directory on `sys.path`, so with `PYTHONPATH=<this repo>:<reference checkout>` a submodule found here
wins and one that is not is looked up there.  `ps_vae.utils` falls through name by name (see utils.py)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
