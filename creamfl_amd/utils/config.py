"""Two-level attribute config (stand-in for munch + src/utils/config.py:102 parse_config).

`Config` is a dict with attribute access and `.get`, which is all the hot path reads
(`config.model.embed_dim`, `config.criterion.get('uniform_lambda', 0)` ...).  `parse_config` reads
the same yaml layout as src/coco.yaml / src/f30k.yaml; `default_config()` returns the values of
src/coco.yaml that matter for the hot path so nothing has to be read from the reference tree.
"""
import copy

import yaml


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return Config(copy.deepcopy(dict(self)))


def _wrap(d):
    if isinstance(d, dict):
        return Config({k: _wrap(v) for k, v in d.items()})
    return d


def parse_config(config_fname, strict_cast=True, verbose=False, **kwargs):
    """src/utils/config.py:102 -- yaml -> attribute dict; `a__b=v` kwargs override config.a.b."""
    with open(config_fname) as f:
        cfg = _wrap(yaml.safe_load(f))
    for arg_key, arg_val in kwargs.items():
        keys = arg_key.split('__')
        if len(keys) != 2:
            raise ValueError(f'invalid override key {arg_key}')
        cfg[keys[0]][keys[1]] = arg_val
    return cfg


def default_config(embed_dim=256, cnn_type='resnet101', not_bert=False):
    """The hot-path-relevant values of src/coco.yaml (model/optimizer/criterion/train sections)."""
    return _wrap({
        'dataloader': {'batch_size': 128, 'eval_batch_size': 8, 'crop_size': 224, 'word_dim': 300},
        'model': {'name': 'pcme', 'embed_dim': embed_dim, 'cnn_type': cnn_type, 'wemb_type': None,
                  'word_dim': 300, 'cache_dir': None, 'n_samples_inference': 7, 'eval_method': 'matmul',
                  'not_bert': not_bert, 'use_img_client': True, 'use_txt_client': True, 'use_mm_client': True},
        'optimizer': {'name': 'adamp', 'learning_rate': 0.0002, 'weight_decay': 0.0},
        'lr_scheduler': {'name': 'cosine_annealing', 'T_max': 30},
        'criterion': {'name': 'pcme', 'init_negative_scale': 15, 'init_shift': 15, 'num_samples': 7, 'vib_beta': 0},
        'train': {'grad_clip': 2, 'use_fp16': True, 'log_step': 100, 'output_file': 'model_noprob.log'},
    })
