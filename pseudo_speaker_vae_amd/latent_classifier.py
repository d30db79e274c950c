"""LatentClassifier: the reference's MLP head on latent vectors (/root/reference/ps_vae/latent_classifier.py:7-70).

Same constructor arguments, module layout (`layers` ModuleList with activations interleaved,
`output_layers` ModuleDict) and therefore the same state_dict keys.  It is tiny (64 -> C); SURVEY.md
section 8f ranks it "next" (N1): it runs on plain torch ops here, the HIP path covers the VAE itself.
"""
from __future__ import annotations

import torch.nn as nn

_ACTIVATIONS = {"relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "leaky_relu": nn.LeakyReLU}


class LatentClassifier(nn.Module):
    def __init__(self, input_dim: int, num_classes, num_layers: int = 1, hidden_dim: int = 128,
                 activation: str = "relu"):
        super().__init__()
        if activation not in _ACTIVATIONS:
            raise ValueError(f"Unsupported activation: {activation}")
        act = _ACTIVATIONS[activation]
        self.single_label_mode = isinstance(num_classes, int)
        self.num_classes = num_classes
        self.layers = nn.ModuleList()
        self.output_layers = nn.ModuleDict()
        heads = None if self.single_label_mode else dict(num_classes)
        if num_layers == 1:
            if self.single_label_mode:
                self.layers.append(nn.Linear(input_dim, num_classes))
            else:
                for label, n in heads.items():
                    self.output_layers[label] = nn.Linear(input_dim, n)
            return
        self.layers.append(nn.Linear(input_dim, hidden_dim))
        for _ in range(num_layers - 2):
            self.layers.append(act())
            self.layers.append(nn.Linear(hidden_dim, hidden_dim))
        self.layers.append(act())
        if self.single_label_mode:
            self.layers.append(nn.Linear(hidden_dim, num_classes))
        else:
            for label, n in heads.items():
                self.output_layers[label] = nn.Linear(hidden_dim, n)

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        if self.single_label_mode:
            return x
        return {label: head(x) for label, head in self.output_layers.items()}
