"""Mirror of src/losses/__init__.py:11-38.  Only 'softmax' -> nn.CrossEntropyLoss is ever selected by
CreamFL (ClientTrainer default loss='softmax', ClientTrainer.py:137,280); the legacy torch-0.3
metric-learning losses of the reference are dead code and are not provided."""
import torch.nn as nn

__factory = {
    'softmax': nn.CrossEntropyLoss,
}


def names():
    return sorted(__factory.keys())


def create(name, *args, **kwargs):
    """Create a loss instance."""
    if name not in __factory:
        raise KeyError("Unknown loss:", name)
    return __factory[name](*args, **kwargs)
