"""ps_vae.embedding_classifier: `embedding_classifier.EmbeddingClassifier` (the consistency head,
/root/reference/ps_vae/lightning.py:4,44-52) resolves to pseudo_speaker_vae_amd.embedding_classifier; the head's own
training driver (`train_embedding_classifier.py`, out of scope) keeps resolving to the reference checkout."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
