from pseudo_speaker_vae_amd.model import PsVAE, VAEModel  # noqa: F401
