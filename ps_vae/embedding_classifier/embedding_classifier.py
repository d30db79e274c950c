from pseudo_speaker_vae_amd.embedding_classifier import EmbeddingClassifier  # noqa: F401
