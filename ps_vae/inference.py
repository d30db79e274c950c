from pseudo_speaker_vae_amd.sampling import conditional_synthesis, unconditional_synthesis  # noqa: F401
