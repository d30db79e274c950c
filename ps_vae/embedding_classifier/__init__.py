"""ps_vae.embedding_classifier: resolves to pseudo_speaker_vae_amd.embedding_classifier (reference: ps_vae/embedding_classifier/)."""
