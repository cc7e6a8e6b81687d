"""Multi-modal client engine base.  Mirrors src/algorithms/base.py:62-230 (EngineBase): builds PCME (ResNet-18 +
GRU text tower), MCSoftContrastiveLoss, the optimizer / scheduler and an evaluator from a two-level config.
Loaders are injected (`train_loader`, `val_loader`): the Flickr30k dataset classes are out of scope."""
import torch

from .. import runtime
from ..criterions import get_criterion
from ..networks.models import get_model
from .eval_coco import COCOEvaluator
from .optimizers import get_lr_scheduler, get_optimizer


class EngineBase(object):
    def __init__(self, args, config, logger, client=-1, dset_name="flicker30k", device='cuda',
                 vocab_path='./datasets/vocabs/coco_vocab.pkl', mlp_local=False, word2idx=None, train_loader=None,
                 val_loader=None):
        runtime.configure()                  # same library set-up as the server engine (creamfl_amd/runtime.py)
        self.dset_name = dset_name
        self.args = args
        self.config = config
        self.device = device
        self.evaluator = COCOEvaluator(eval_method=config.model.get('eval_method', 'matmul'), verbose=False,
                                       eval_device=device, extract_device=device, n_crossfolds=5)
        self.logger = logger
        self.metadata = {}
        self.client = client
        self.train_loader, self.val_loader = train_loader, val_loader
        if word2idx is None:
            from ..networks.language_model import COCO_VOCAB_SIZE
            word2idx = {i: i for i in range(COCO_VOCAB_SIZE)}
        self.word2idx = word2idx
        self.model = get_model(word2idx, config.model, mlp_local)
        self.set_criterion(get_criterion(config.criterion.name, config.criterion))
        params = [p for p in self.model.parameters() if p.requires_grad]
        params += [p for p in self.criterion.parameters() if p.requires_grad]
        self.set_optimizer(get_optimizer(config.optimizer.name, params, config.optimizer))
        self.set_lr_scheduler(get_lr_scheduler(config.lr_scheduler.name, self.optimizer, config.lr_scheduler))
        self.evaluator.set_model(self.model)
        self.evaluator.set_criterion(self.criterion)
        self.cur_epoch = 0
        self.old_model = None
        self.local_epochs = args.local_epochs
        self.local_epoch = 0
        self.autocast_dtype = None

    def model_to_device(self):
        self.model.to(self.device)
        if self.criterion:
            self.criterion.to(self.device)

    def set_optimizer(self, optimizer):
        self.optimizer = optimizer

    def set_criterion(self, criterion):
        self.criterion = criterion

    def set_lr_scheduler(self, lr_scheduler):
        self.lr_scheduler = lr_scheduler

    def set_evaluator(self, evaluator):
        self.evaluator = evaluator

    def to_half(self):
        """bf16 autocast + channels_last instead of apex amp O2 (base.py:143-147)."""
        self.autocast_dtype = torch.bfloat16
        self.model.to(memory_format=torch.channels_last)
        self.evaluator.autocast_dtype = self.autocast_dtype

    @torch.no_grad()
    def evaluate(self, val_loaders, n_crossfolds=None, **kwargs):
        self.model_to_device()
        self.model.eval()
        if not isinstance(val_loaders, dict):
            val_loaders = {'te': val_loaders}
        scores = {}
        n_crossfolds = self.evaluator.n_crossfolds if n_crossfolds is None else n_crossfolds
        for key, data_loader in val_loaders.items():
            _n = -1 if key == 'val' else n_crossfolds
            scores[key] = self.evaluator.evaluate(
                data_loader, n_crossfolds=_n, key=key,
                n_images_per_crossfold=int(data_loader.dataset.n_images / _n),
                n_captions_per_crossfold=int(len(data_loader.dataset) / _n), **kwargs)
        return scores

    def save_models(self, save_to, metadata=None):
        torch.save({'model': self.model.state_dict(), 'config': dict(self.config), 'metadata': metadata}, save_to)

    def load_models(self, state_dict_path, load_keys=None):
        state_dict = torch.load(state_dict_path, map_location='cpu')
        self.model.load_state_dict(state_dict.get('model', state_dict), strict=False)
