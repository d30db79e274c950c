from pseudo_speaker_vae_amd.latent_classifier import LatentClassifier  # noqa: F401
