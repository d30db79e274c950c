from pseudo_speaker_vae_amd.lightning import PseudoSpeakerVAE  # noqa: F401
