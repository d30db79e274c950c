"""`ps_vae.utils`: the label maps and the config loader (/root/reference/ps_vae/utils.py:69-136) come from the
implementation; every other name of the reference's module (`LatentSpacePCACallback`, utils.py:12-67;
`sample_from_cv_metadata`, utils.py:138-167; anything added later) falls through to the reference's own
`ps_vae/utils.py`, found further along `ps_vae.__path__` and loaded on first use -- not a closed list."""
import importlib.util as _ilu
import os as _os
import sys as _sys

from pseudo_speaker_vae_amd.labels import (METADATA_TRANSFORMS, load_yaml_config, map_cv_age_to_label,  # noqa: F401
                                           map_cv_gender_to_label, map_vctk_gender_to_label)

_REF_NAME = "ps_vae._reference_utils"


def _reference_utils_path():
    """The next `ps_vae/utils.py` after this one on the package path, or None."""
    import ps_vae
    here = _os.path.realpath(__file__)
    for d in ps_vae.__path__:
        cand = _os.path.join(d, "utils.py")
        if _os.path.isfile(cand) and _os.path.realpath(cand) != here:
            return cand
    return None


def _reference_utils():
    mod = _sys.modules.get(_REF_NAME)
    if mod is None:
        path = _reference_utils_path()
        if path is None:
            return None
        spec = _ilu.spec_from_file_location(_REF_NAME, path)
        mod = _ilu.module_from_spec(spec)
        _sys.modules[_REF_NAME] = mod
        try:
            spec.loader.exec_module(mod)
        except BaseException:
            _sys.modules.pop(_REF_NAME, None)
            raise
    return mod


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    try:
        ref = _reference_utils()
    except ImportError as e:  # the reference's module needs pytorch_lightning, torchvision, matplotlib, sklearn
        raise ImportError(f"ps_vae.utils.{name} lives in the reference's ps_vae/utils.py "
                          f"({_reference_utils_path()}), which failed to import: {e}") from e
    if ref is None:
        raise AttributeError(f"module 'ps_vae.utils' has no attribute {name!r} (not part of the MI355X hot path, and "
                             f"no reference checkout follows this repo on sys.path to take it from)")
    return getattr(ref, name)
