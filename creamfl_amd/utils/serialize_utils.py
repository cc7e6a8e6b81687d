"""Mirror of src/utils/serialize_utils.py:9 flatten_dict (no pandas)."""


def flatten_dict(d, sep='_', prefix=''):
    out = {}
    for k, v in d.items():
        key = f'{prefix}{sep}{k}' if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten_dict(v, sep, key))
        else:
            out[key] = v
    return out
