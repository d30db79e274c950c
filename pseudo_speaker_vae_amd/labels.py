"""Label -> index maps of the reference (/root/reference/ps_vae/utils.py:82-136) and the combined
metadata transform of /root/reference/ps_vae/data/cv.py:8-15.  These integers are the "bit-exact
latent indices" of the conditional one-hot extension."""
from __future__ import annotations

_AGE = {"teens": 0, "twenties": 0, "thirties": 1, "fourties": 1, "fifties": 1,
        "sixties": 2, "seventies": 2, "eighties": 2, "nineties": 2}
_CV_GENDER = {"male": 0, "female": 1, "other": 2}
_VCTK_GENDER = {"M": 0, "F": 1}


def map_cv_age_to_label(age) -> int:
    """utils.py:82-103: three age bands, -1 when unknown."""
    return _AGE.get(age, -1)


def map_cv_gender_to_label(gender) -> int:
    """utils.py:105-120."""
    return _CV_GENDER.get(gender, -1)


def map_vctk_gender_to_label(gender) -> int:
    """utils.py:122-136."""
    return _VCTK_GENDER.get(gender, -1)


METADATA_TRANSFORMS = {  # data/cv.py:8-15
    "gender": lambda x: map_cv_gender_to_label(x["gender"]),
    "age": lambda x: map_cv_age_to_label(x["age"]),
    "age_and_gender": lambda x: {"age": map_cv_age_to_label(x["age"]), "gender": map_cv_gender_to_label(x["gender"])},
}


def load_yaml_config(file_path: str) -> dict:
    """utils.py:69-80."""
    import yaml
    with open(file_path, "r") as f:
        return yaml.safe_load(f)
