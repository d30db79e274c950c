from pseudo_speaker_vae_amd.labels import (METADATA_TRANSFORMS, load_yaml_config, map_cv_age_to_label,  # noqa: F401
                                           map_cv_gender_to_label, map_vctk_gender_to_label)
