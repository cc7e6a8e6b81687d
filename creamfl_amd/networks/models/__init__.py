"""Mirror of src/networks/models/__init__.py:1-6."""
__all__ = ['get_model']

from .pcme import PCME


def get_model(word2idx, config, mlp_local):
    return PCME(word2idx, config, mlp_local)
